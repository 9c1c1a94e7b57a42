"""Measurement only: A/B timing of conv3 fwd / wgrad at the V-Net layer shapes across library variants (the product library
plus every tools/_abl/libbcp_abl_*.so built by tools/ablate_conv.sh or by hand with -D switches).  Variants are INTERLEAVED
round-robin over several rounds and the median per variant is reported: back-to-back blocks of one variant each were biased by
clock / thermal drift by +-5 %.   python tools/ablate_conv.py [C ...]"""
import glob
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bcp_amd import _lib  # noqa: E402
from bcp_amd.hip_ops import Ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = {"16": (16, (112, 112, 80), 1), "16x2": (16, (112, 112, 80), 2), "32": (32, (56, 56, 40), 2), "64": (64, (28, 28, 20), 2),
          "128": (128, (14, 14, 10), 2), "256": (256, (7, 7, 5), 2),
          # ACDC 2-D U-Net levels (grouped batch of 12 slices)
          "2d16": (16, (1, 256, 256), 12), "2d32": (32, (1, 128, 128), 12), "2d64": (64, (1, 64, 64), 12), "2d128": (128, (1, 32, 32), 12),
          "2d256": (256, (1, 16, 16), 12)}
which = sys.argv[1:] or ["16x2", "32"]
libs = [("product", _lib.LIB_PATH)] + sorted((os.path.basename(p)[11:-3], p) for p in glob.glob(os.path.join(ROOT, "tools", "_abl", "libbcp_abl_*.so")))
ROUNDS, ITERS = 7, 10
for wname in which:
    C, sp, N = shapes[wname]
    KD = 1 if sp[0] == 1 and wname.startswith("2d") else 3
    x = torch.randn(N, *sp, C, device=dev)
    dy = torch.randn(N, *sp, C, device=dev)
    w = torch.randn(C, C, *((3, 3) if KD == 1 else (3, 3, 3)), device=dev) * 0.05
    b = torch.zeros(C, device=dev)
    y = torch.empty(N, *sp, C, device=dev)
    dw = torch.empty_like(w)
    flops = 2.0 * N * sp[0] * sp[1] * sp[2] * 9 * KD * C * C
    opsl = []
    for name, path in libs:
        ops = Ops(_lib.Binding(path), allow_cpu=False)
        wf, _ = ops.conv3_pack(w, KD)
        opsl.append((name, ops, wf, ops.event(), ops.event()))
        for _ in range(3):
            ops.conv3_fwd(x, wf, b, C, KD, out=y)
            ops.conv3_wgrad(x, dy, dw, KD)
    tf, tw = {n: [] for n, *_ in opsl}, {n: [] for n, *_ in opsl}
    for r in range(ROUNDS):
        for name, ops, wf, e0, e1 in opsl:
            ops.event_record(e0, x)
            for _ in range(ITERS):
                ops.conv3_fwd(x, wf, b, C, KD, out=y)
            ops.event_record(e1, x)
            tf[name].append(ops.event_elapsed_ms(e0, e1) / ITERS)
            ops.event_record(e0, x)
            for _ in range(ITERS):
                ops.conv3_wgrad(x, dy, dw, KD)
            ops.event_record(e1, x)
            tw[name].append(ops.event_elapsed_ms(e0, e1) / ITERS)
    for name, *_ in opsl:
        mf, mw = statistics.median(tf[name]), statistics.median(tw[name])
        print(f"C={C:3d} N={N} {name:10s} fwd {mf * 1e3:7.1f} us {flops / mf / 1e9:6.1f} TF/s (min {min(tf[name]) * 1e3:6.1f}) | "
              f"wgrad {mw * 1e3:7.1f} us {flops / mw / 1e9:6.1f} TF/s (min {min(tw[name]) * 1e3:6.1f})", flush=True)
