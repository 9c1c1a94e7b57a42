"""Measurement only: time conv3 fwd at the V-Net layer shapes with each ablated library built by tools/ablate_conv.sh.
   python tools/ablate_conv.py [C ...]"""
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bcp_amd import _lib  # noqa: E402
from bcp_amd.hip_ops import Ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = {"16": (16, (112, 112, 80), 1), "32": (32, (56, 56, 40), 2), "64": (64, (28, 28, 20), 2), "128": (128, (14, 14, 10), 2), "256": (256, (7, 7, 5), 2)}
which = sys.argv[1:] or ["16", "32"]
libs = [("product", _lib.LIB_PATH)] + sorted((os.path.basename(p)[11:-3], p) for p in glob.glob(os.path.join(ROOT, "tools", "_abl", "libbcp_abl_*.so")))
for wname in which:
    C, sp, N = shapes[wname]
    x = torch.randn(N, *sp, C, device=dev)
    w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
    b = torch.zeros(C, device=dev)
    y = torch.empty(N, *sp, C, device=dev)
    flops = 2.0 * N * sp[0] * sp[1] * sp[2] * 27 * C * C
    for name, path in libs:
        ops = Ops(_lib.Binding(path), allow_cpu=False)
        wf, _ = ops.conv3_pack(w, 3)
        for _ in range(3):
            ops.conv3_fwd(x, wf, b, C, 3, out=y)
        e0, e1 = ops.event(), ops.event()
        ops.event_record(e0, x)
        for _ in range(20):
            ops.conv3_fwd(x, wf, b, C, 3, out=y)
        ops.event_record(e1, x)
        ms = ops.event_elapsed_ms(e0, e1) / 20
        dy = torch.randn(N, *sp, C, device=dev)
        dw = torch.empty_like(w)
        for _ in range(3):
            ops.conv3_wgrad(x, dy, dw, 3)
        ops.event_record(e0, x)
        for _ in range(20):
            ops.conv3_wgrad(x, dy, dw, 3)
        ops.event_record(e1, x)
        msw = ops.event_elapsed_ms(e0, e1) / 20
        print(f"C={C:3d} N={N} abl={name:8s} fwd {ms * 1e3:8.1f} us {flops / ms / 1e9:7.1f} TF/s | wgrad {msw * 1e3:8.1f} us {flops / msw / 1e9:7.1f} TF/s", flush=True)
