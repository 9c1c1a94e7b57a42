"""round 6, measurement only: the U-Net's bilinear x2 backward / max-pool backward alone at the ACDC step's four levels (batch 12), HIP events
back to back, with the bytes each must move.   python tools/probe/pool_probe.py [name=value ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bcp_amd.hip_ops import Ops  # noqa: E402

ops = Ops.product()
dev = torch.device("cuda:0")
for a in sys.argv[1:]:
    k, _, v = a.partition("=")
    ops.set_option(k, int(v))


def timeit(fn, like, iters=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = ops.event(), ops.event()
    ops.event_record(e0, like)
    for _ in range(iters):
        fn()
    ops.event_record(e1, like)
    return ops.event_elapsed_ms(e0, e1) / iters * 1e3


N = 12
for C, HW in ((16, 128), (32, 64), (64, 32), (128, 16)):          # C = channels of the upsampled half, HW = the COARSE extent
    dcat = torch.randn(N, 1, 2 * HW, 2 * HW, 2 * C, device=dev)
    t = timeit(lambda: ops.bilinear2x_bwd(dcat, C, C), dcat)
    mb = (N * 4 * HW * HW * C + N * HW * HW * C) * 4 / 1e6
    print(f"RESULT bilinear2x_bwd {N}x{2 * HW}^2 x{C} -> {HW}^2: {t:6.1f} us  {mb:6.1f} MB  {mb / t * 1e-6 * 1e6 / 1e6 * 1e3:6.0f} GB/s", flush=True)
    x = torch.randn(N, 1, 2 * HW, 2 * HW, C, device=dev)
    dy = torch.randn(N, 1, HW, HW, C, device=dev)
    dx = torch.empty_like(x)
    add = torch.randn(N, 1, 2 * HW, 2 * HW, 2 * C, device=dev)
    t = timeit(lambda: ops.maxpool2d_bwd(x, dy, dx, add=ops.channel_slab(add, C)), x)
    mb = (3 * N * 4 * HW * HW * C + N * HW * HW * C) * 4 / 1e6
    print(f"RESULT maxpool2d_bwd  {N}x{2 * HW}^2 x{C} <- {HW}^2: {t:6.1f} us  {mb:6.1f} MB  {mb / t * 1e3:6.0f} GB/s", flush=True)
