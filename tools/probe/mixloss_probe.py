"""round 6, measurement only: bcp_mixloss_fwd / bcp_mixloss_pair_fwd alone at the LA and ACDC step sizes (HIP events, back to back).
   python tools/probe/mixloss_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bcp_amd import hip_ops as H  # noqa: E402
from bcp_amd.hip_ops import Ops  # noqa: E402

ops = Ops.product()
dev = torch.device("cuda:0")


def timeit(fn, like, iters=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = ops.event(), ops.event()
    ops.event_record(e0, like)
    for _ in range(iters):
        fn()
    ops.event_record(e1, like)
    return ops.event_elapsed_ms(e0, e1) / iters * 1e3


g = torch.Generator().manual_seed(0)
for name, shape, C, flav, box in (("LA", (2, 112, 112, 80), 2, H.LOSS_LA, (20, 30, 10, 74, 74, 53)), ("ACDC", (12, 1, 256, 256), 4, H.LOSS_ACDC, (0, 40, 60, 1, 170, 170))):
    lo = torch.randn(*shape, C, generator=g).to(dev)
    lo2 = torch.randn(2 * shape[0], *shape[1:], C, generator=g).to(dev)
    la = torch.randint(0, C, shape, generator=g).to(torch.uint8).to(dev)
    lb = torch.randint(0, C, shape, generator=g).to(torch.uint8).to(dev)
    t1 = timeit(lambda: ops.mixloss_fwd(lo, la, lb, box, flav, 1.0, 0.5), lo)
    t2 = timeit(lambda: ops.mixloss_pair_fwd(lo2, la, lb, lb, la, box, flav, (1.0, 0.5), (0.5, 1.0)), lo)
    vox = lo.numel() // C
    print(f"RESULT {name}: mixloss_fwd {t1:6.1f} us ({vox * (4 * C + 2) / t1 * 1e-6:5.2f} TB/s)   pair_fwd {t2:6.1f} us ({2 * vox * (4 * C + 2) / t2 * 1e-6:5.2f} TB/s)", flush=True)
