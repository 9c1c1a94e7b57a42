"""round 6: k2s2 / transposed conv forward + norm with and without the statistics in the GEMM epilogue, at the LA V-Net's four levels
(HIP events, back to back).   python tools/probe/k2_stats_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bcp_amd import hip_ops as H  # noqa: E402
from bcp_amd.hip_ops import Ops  # noqa: E402

ops = Ops.product()
dev = torch.device("cuda:0")
for a in sys.argv[1:]:                 # library options: name=value  (e.g. gemm_pipe=0 gemm_stat_r=4)
    k, _, v = a.partition("=")
    ops.set_option(k, int(v))
print("OPTIONS", sys.argv[1:], flush=True)


def timeit(fn, like, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = ops.event(), ops.event()
    ops.event_record(e0, like)
    for _ in range(iters):
        fn()
    ops.event_record(e1, like)
    return ops.event_elapsed_ms(e0, e1) / iters * 1e3


N, G = 2, 2
for kind, Cin, Cout, sp in ((1, 32, 16, (56, 56, 40)), (1, 64, 32, (28, 28, 20)), (0, 16, 32, (112, 112, 80)), (0, 32, 64, (56, 56, 40)), (1, 128, 64, (14, 14, 10))):
    x = torch.randn(N, *sp, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 2, 2, 2, device=dev) if kind == 0 else torch.randn(Cin, Cout, 2, 2, 2, device=dev)) * 0.1
    bp = ops.k2_pack(w, Cin, Cout, H.PACK_DOWN_FWD if kind == 0 else H.PACK_UP_FWD)
    b = torch.zeros(Cout, device=dev)
    g, be, rm, rv = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
    plain = ops.down_fwd if kind == 0 else ops.up_fwd
    y = plain(x, bp, b, Cout)
    a = torch.empty_like(y)
    rows = ops.k2_stat_rows(kind, x.shape, Cout, G)
    t_g = timeit(lambda: plain(x, bp, b, Cout, out=y), x)
    t_n = timeit(lambda: ops.norm_fwd(y, G, g, be, rm, rv, H.ACT_RELU, out=a), x)
    t_c = timeit(lambda: ops.norm_fwd(plain(x, bp, b, Cout, out=y), G, g, be, rm, rv, H.ACT_RELU, out=a), x)
    line = f"RESULT kind {kind} {Cin}->{Cout} x{sp}: gemm {t_g:6.1f}  norm {t_n:6.1f}  chain {t_c:6.1f} us"
    if rows:
        def fused():
            yy, part, nb = ops.k2_fwd_stats(kind, x, bp, b, Cout, G)
            ops.norm_fwd(yy, G, g, be, rm, rv, H.ACT_RELU, out=a, partial=part, nb=nb)
        t_s = timeit(lambda: ops.k2_fwd_stats(kind, x, bp, b, Cout, G), x)
        yy, part, nb = ops.k2_fwd_stats(kind, x, bp, b, Cout, G)
        t_np = timeit(lambda: ops.norm_fwd(yy, G, g, be, rm, rv, H.ACT_RELU, out=a, partial=part, nb=nb), x)
        t_f = timeit(fused, x)
        line += f"   | rows {rows}: gemm+stats {t_s:6.1f}  norm(partial) {t_np:6.1f}  chain {t_f:6.1f} us"
    print(line, flush=True)
