"""measurement: the chaotic-trajectory and gradient checks under the two-plane fp16 conv instances (conv3_f16 = 1) and the three-plane
bf16 ones (0), numbers side by side (round 4).  Run on the GPU box from the repo root: python tools/diag/f16_parity_diag.py"""
import os
import sys

sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import numpy as np
import torch
import net_checks as NC
from bcp_amd.hip_ops import Ops

ops = Ops.product()
dev = torch.device("cuda:0")
G = "tests/golden"
for f16 in (1, 0):
    ops.set_option("conv3_f16", f16)
    tag = "f16x2" if f16 else "bf16x3"
    rep = []
    try:
        NC.check_la_traj5(ops, dev, G, report=rep, fixture="la_traj5f.npz")
    except AssertionError as e:
        print(tag, "la_traj5f ASSERT", str(e)[:160])
    for r in rep:
        print(tag, "la_traj5f step %d: |hip-ref32| %.2e |hip-ref64f| %.2e (ens median %.2e) plab %d (ref %d)" % r)
    g = np.load(os.path.join(G, "la_traj5f.npz"))
    print(tag, "  ensemble max per step", ["%.2e" % v for v in g["drift_ens"].max(axis=0)])
    rep = []
    try:
        NC.check_acdc_traj5(ops, dev, G, report=rep, fixture="acdc_traj5f.npz", floor=2e-5, factor=4.0)
    except AssertionError as e:
        print(tag, "acdc_traj5f ASSERT", str(e)[:160])
    for r in rep:
        print(tag, "acdc_traj5f step %d: |hip-ref32| %.2e |hip-ref64| %.2e (ref 32 vs 64 %.2e) plab diff %.0f (ref %.0f)" % r)
    try:
        NC.check_la_unfused_loop(ops, dev, G, steps=3)
        print(tag, "la_unfused_loop ok")
    except AssertionError as e:
        print(tag, "la_unfused_loop ASSERT", str(e)[:300])
    try:
        print(tag, "pattern grads la 112x112x80", NC.check_vnet_pattern_grads(ops, dev, "la", (112, 112, 80), seed=31, N=1))
    except AssertionError as e:
        print(tag, "pattern grads ASSERT", str(e)[:300])
    rep = {}
    try:
        NC.check_la_step_full(ops, dev, report=rep)
    except AssertionError as e:
        print(tag, "la_step_full ASSERT", str(e)[:300])
    print(tag, "full-size step:", rep)
