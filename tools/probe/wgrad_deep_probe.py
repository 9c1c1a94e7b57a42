"""round 6: the deep-level weight gradient (csrc/conv3bw.hip k_w6) as rounds 2-5 launched it (512 slots of partial slabs + reduce) against the
round-6 launches (few tile groups; one group = the kernel writes dW itself): time per call (HIP events, back to back) and the difference of
the results.   python tools/probe/wgrad_deep_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bcp_amd import hip_ops as H  # noqa: E402
from bcp_amd.hip_ops import Ops  # noqa: E402

ops = Ops.product()
dev = torch.device("cuda:0")


def timeit(fn, like, iters=30, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = ops.event(), ops.event()
    ops.event_record(e0, like)
    for _ in range(iters):
        fn()
    ops.event_record(e1, like)
    return ops.event_elapsed_ms(e0, e1) / iters * 1e3


SHAPES = [(2, (7, 7, 5), 256), (2, (14, 14, 10), 128), (2, (6, 6, 6), 256), (2, (12, 12, 12), 128)]
SETS = [("r05", {"wgrad_b6_deep": 0})]
for tile in (1, 0):
    for nt in (1, 2):
        for slots in (128, 256, 512):
            SETS.append((f"deep t{tile} nt{nt} s{slots}", {"wgrad_b6_deep": 1, "wgrad_b6_deep_nt": nt, "wgrad_b6_deep_slots": slots, "wgrad_b6_deep_tile": tile}))
for N, sp, C in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(N, *sp, C, device=dev)
    dy = torch.randn(N, *sp, C, device=dev) * 1e-3
    x._bcp_amax = H.amax_slots(float(x.abs().max()), dev)
    dy._bcp_amax = H.amax_slots(float(dy.abs().max()), dev)
    ref = None
    for name, opts in SETS:
        for k, v in opts.items():
            ops.set_option(k, v)
        dw = torch.zeros(C, C, 3, 3, 3, device=dev)
        us = timeit(lambda: ops.conv3_wgrad(x, dy, dw, 3), x)
        torch.cuda.synchronize()
        if ref is None:
            ref = dw.clone()
        err = float((dw - ref).abs().max() / ref.abs().max())
        ops.conv3_wgrad(x, dy, dw, 3, accumulate=True)
        torch.cuda.synchronize()
        err2 = float((dw - 2 * ref).abs().max() / ref.abs().max())
        print(f"RESULT {N}x{sp}x{C:4d} {name:30s} {us:7.1f} us   max|d|/max|ref| {err:.2e}  (+=: {err2:.2e})", flush=True)
        for k in opts:
            ops.set_option(k)
