"""norm_fwd / norm_bwd at the LA and ACDC in-step shapes, idle GPU, per option value of stats_il (csrc/common.h): time per call by an event
pair around 20 back-to-back calls (the statistics pass is one of the call's three kernels; differences between option values are its).
  (needs tools/attic/stats_il.patch applied: the option is not in the tree)  python tools/attic/stats_pass_bench.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from bcp_amd.hip_ops import Ops
import bcp_amd.hip_ops as H
ops = Ops.product(); dev = torch.device("cuda:0")
shapes = [((2, 112, 112, 80, 16), 1, True), ((2, 56, 56, 40, 32), 1, False), ((2, 28, 28, 20, 64), 1, False), ((24, 1, 256, 256, 16), 2, False), ((24, 1, 128, 128, 32), 2, False), ((24, 1, 64, 64, 64), 2, False)]
for shape, G, drop in shapes:
    C = shape[-1]
    y = torch.randn(*shape, device=dev) * 1.3 + 0.2
    da = torch.randn(*shape, device=dev) * 1e-3
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.2
    cs = (torch.randint(0, 2, (shape[0], C), device=dev).float() * 2) if drop else None
    ref = None
    for il in (0, 1, 2, 3):
        ops.set_option("stats_il", il)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        a, st = ops.norm_fwd(y, G, gam, bet, rm, rv, H.ACT_RELU, chan_scale=cs)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dy = ops.norm_bwd(y, da, G, st, H.ACT_RELU, dg, db, False, chan_scale=cs)
        torch.cuda.synchronize()
        if ref is None:
            ref = (st.clone(), dy.clone(), dg.clone())
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        for _ in range(20):
            ops.norm_fwd(y, G, gam, bet, rm, rv, H.ACT_RELU, chan_scale=cs)
        e[1].record(); e[2].record()
        for _ in range(20):
            ops.norm_bwd(y, da, G, st, H.ACT_RELU, dg, db, False, chan_scale=cs, out=dy)
        e[3].record(); torch.cuda.synchronize()
        rel = lambda p, q: float((p - q).abs().max()) / max(float(q.abs().max()), 1e-30)
        print(f"{shape} G={G} stats_il={il}: norm_fwd {e[0].elapsed_time(e[1]) * 50:.1f} us, norm_bwd {e[2].elapsed_time(e[3]) * 50:.1f} us; vs stats_il=0: stats {rel(st, ref[0]):.1e} dy {rel(dy, ref[1]):.1e} dgamma {rel(dg, ref[2]):.1e}", flush=True)
    ops.set_option("stats_il")
