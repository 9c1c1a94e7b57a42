"""Run a few launches of the dominant conv kernels for rocprofv3 --pmc passes.  python tools/prof_conv.py [C]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bcp_amd.hip_ops import Ops  # noqa: E402

ops = Ops.product()
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "16"
shapes = {"16": (16, (112, 112, 80)), "32": (32, (56, 56, 40)), "64": (64, (28, 28, 20)), "128": (128, (14, 14, 10)), "256": (256, (7, 7, 5))}
C, sp = shapes[which]
NB = 2   # the in-step launch shape: grouped batch of 2
x = torch.randn(NB, *sp, C, device=dev)
dy = torch.randn(NB, *sp, C, device=dev)
w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
b = torch.zeros(C, device=dev)
wf, wd = ops.conv3_pack(w, 3)
y = torch.empty(NB, *sp, C, device=dev)
dw = torch.empty_like(w)
for _ in range(5):
    ops.conv3_fwd(x, wf, b, C, 3, out=y)
for _ in range(5):
    ops.conv3_wgrad(x, dy, dw, 3)
torch.cuda.synchronize()
