"""time plain dgrad vs dgrad + fused norm-backward statistics vs the standalone statistics pass, at the in-step batch of 2"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bcp_amd import hip_ops as H
from bcp_amd.hip_ops import Ops
ops = Ops.product(); dev = torch.device("cuda:0")
def timeit(fn, like, iters=20):
    for _ in range(3): fn()
    e0, e1 = ops.event(), ops.event()
    ops.event_record(e0, like)
    for _ in range(iters): fn()
    ops.event_record(e1, like)
    return ops.event_elapsed_ms(e0, e1) / iters * 1e3
for C, sp in ((16, (112, 112, 80)), (32, (56, 56, 40)), (64, (28, 28, 20)), (128, (14, 14, 10))):
    N, G = 2, 2
    y = torch.randn(N, *sp, C, device=dev); dy = torch.randn(N, *sp, C, device=dev)
    w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
    g, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    _, st = ops.norm_fwd(y, G, g, be, torch.zeros(C, device=dev), torch.ones(C, device=dev), H.ACT_RELU)
    _, wd = ops.conv3_pack(w, 3)
    t0 = timeit(lambda: ops.conv3_fwd(dy, wd, None, C, 3), y)
    t1 = timeit(lambda: ops.conv3_dgrad_bwdstats(dy, wd, C, 3, y, st, H.ACT_RELU, G), y)
    da = ops.conv3_fwd(dy, wd, None, C, 3)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    out = torch.empty_like(y)
    t2 = timeit(lambda: ops.norm_bwd(y, da, G, st, H.ACT_RELU, dg, db, False, out=out), y)
    da2, part, rows = ops.conv3_dgrad_bwdstats(dy, wd, C, 3, y, st, H.ACT_RELU, G)
    t3 = timeit(lambda: ops.norm_bwd(y, da, G, st, H.ACT_RELU, dg, db, False, out=out, partial=part, nb=rows), y) if rows else float("nan")
    print(f"C={C:3d}: dgrad {t0:7.1f} us, dgrad+stats {t1:7.1f} us (+{t1 - t0:5.1f}), norm_bwd {t2:6.1f} us, norm_bwd(fused stats) {t3:6.1f} us (-{t2 - t3:5.1f})  rows={rows}", flush=True)
