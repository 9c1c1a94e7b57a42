"""Per-kernel timing at the LA V-Net layer shapes (hip events on the launch stream).  Prints achieved
TFLOP/s for the MFMA convs and GB/s for the HBM-bound streams.  Usage: python tools/bench_kernels.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bcp_amd import hip_ops as H  # noqa: E402
from bcp_amd.hip_ops import Ops  # noqa: E402


def timeit(ops, fn, like, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = ops.event(), ops.event()
    ops.event_record(e0, like)
    for _ in range(iters):
        fn()
    ops.event_record(e1, like)
    return ops.event_elapsed_ms(e0, e1) / iters


def main():
    ops = Ops.product()
    dev = torch.device("cuda:0")
    rows = []
    layers = [(16, 16, (112, 112, 80)), (32, 32, (56, 56, 40)), (64, 64, (28, 28, 20)), (128, 128, (14, 14, 10)), (256, 256, (7, 7, 5))]
    for Cin, Cout, sp in layers:
        x = torch.randn(1, *sp, Cin, device=dev)
        dy = torch.randn(1, *sp, Cout, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.05
        b = torch.zeros(Cout, device=dev)
        wf, wd = ops.conv3_pack(w, 3)
        y = torch.empty(1, *sp, Cout, device=dev)
        dw = torch.empty_like(w)
        flops = 2.0 * sp[0] * sp[1] * sp[2] * 27 * Cin * Cout
        t = timeit(ops, lambda: ops.conv3_fwd(x, wf, b, Cout, 3, out=y), x)
        rows.append(("conv3_fwd", Cin, sp, t, flops / t / 1e9, None))
        t = timeit(ops, lambda: ops.conv3_wgrad(x, dy, dw, 3), x)
        rows.append(("conv3_wgrad", Cin, sp, t, flops / t / 1e9, None))
        byts = (x.numel() + y.numel()) * 4
        g, be = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
        a = torch.empty_like(y)
        st = [None]

        def nf():
            st[0] = ops.norm_fwd(y, 1, g, be, rm, rv, H.ACT_RELU, out=a)[1]
        t = timeit(ops, nf, x)
        rows.append(("norm_fwd", Cout, sp, t, None, y.numel() * 4 * 3 / t / 1e6))
        dg, db = torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev)
        t = timeit(ops, lambda: ops.norm_bwd(y, dy, 1, st[0], H.ACT_RELU, dg, db, out=a), x)
        rows.append(("norm_bwd", Cout, sp, t, None, y.numel() * 4 * 5 / t / 1e6))
    # first layer, k2s2, output conv, loss, mix, cc at the top level
    sp = (112, 112, 80)
    x1 = torch.randn(1, *sp, 1, device=dev)
    w1 = torch.randn(16, 1, 3, 3, 3, device=dev)
    y16 = torch.empty(1, *sp, 16, device=dev)
    t = timeit(ops, lambda: ops.conv3_c1_fwd(x1, w1, None, 3, out=y16), x1)
    rows.append(("conv3_c1_fwd", 1, sp, t, None, (x1.numel() + y16.numel()) * 4 / t / 1e6))
    dw1 = torch.empty_like(w1)
    t = timeit(ops, lambda: ops.conv3_c1_wgrad(x1, y16, dw1, 3), x1)
    rows.append(("conv3_c1_wgrad", 1, sp, t, None, (x1.numel() + y16.numel()) * 4 / t / 1e6))
    wdn = torch.randn(32, 16, 2, 2, 2, device=dev)
    bp = ops.k2_pack(wdn, 16, 32, H.PACK_DOWN_FWD)
    yd = torch.empty(1, 56, 56, 40, 32, device=dev)
    t = timeit(ops, lambda: ops.down_fwd(y16, bp, None, 32, out=yd), x1)
    rows.append(("down_fwd 16->32", 16, sp, t, 2.0 * yd.numel() * 128 / t / 1e9, (y16.numel() + yd.numel()) * 4 / t / 1e6))
    wup = torch.randn(32, 16, 2, 2, 2, device=dev)
    bpu = ops.k2_pack(wup, 32, 16, H.PACK_UP_FWD)
    t = timeit(ops, lambda: ops.up_fwd(yd, bpu, None, 16, out=y16), x1)
    rows.append(("up_fwd 32->16", 32, sp, t, 2.0 * yd.numel() * 128 / t / 1e9, (y16.numel() + yd.numel()) * 4 / t / 1e6))
    dwd = torch.empty_like(wdn)
    t = timeit(ops, lambda: ops.k2_wgrad(y16, yd, dwd, H.WG_DOWN), x1)
    rows.append(("down_wgrad", 16, sp, t, 2.0 * yd.numel() * 128 / t / 1e9, (y16.numel() + yd.numel()) * 4 / t / 1e6))
    wo = torch.randn(2, 16, 1, 1, 1, device=dev)
    lo = torch.empty(1, *sp, 2, device=dev)
    t = timeit(ops, lambda: ops.pw16_fwd(y16, wo, None, 2, out=lo), x1)
    rows.append(("pw16_fwd", 16, sp, t, None, (y16.numel() + lo.numel()) * 4 / t / 1e6))
    la = (torch.rand(1, *sp, device=dev) > 0.9).to(torch.uint8)
    box = (10, 20, 5, 74, 74, 53)
    t = timeit(ops, lambda: ops.mixloss_fwd(lo, la, la, box, H.LOSS_LA, 1.0, 0.5), x1)
    rows.append(("mixloss_fwd", 2, sp, t, None, lo.numel() * 5 / t / 1e6))
    o3, ws = ops.mixloss_fwd(lo, la, la, box, H.LOSS_LA, 1.0, 0.5)
    t = timeit(ops, lambda: ops.mixloss_bwd(lo, la, la, box, H.LOSS_LA, ws, 0.5, 0.5), x1)
    rows.append(("mixloss_bwd", 2, sp, t, None, lo.numel() * 9 / t / 1e6))
    t = timeit(ops, lambda: ops.mix_box(x1, x1, box), x1)
    rows.append(("mix_box", 1, sp, t, None, x1.numel() * 12 / t / 1e6))
    seg = ops.plabel_bin(lo)
    t = timeit(ops, lambda: ops.cc_largest(la, 1, 3), x1)
    rows.append(("cc_largest(10% fg noise)", 1, sp, t, None, None))
    blob = torch.zeros(1, *sp, dtype=torch.uint8, device=dev)
    blob[:, 20:90, 20:90, 10:70] = 1
    t = timeit(ops, lambda: ops.cc_largest(blob, 1, 3), x1)
    rows.append(("cc_largest(one blob 30%)", 1, sp, t, None, None))
    n = 9457318
    p, gq, bu, em = (torch.randn(n, device=dev) for _ in range(4))
    t = timeit(ops, lambda: ops.sgd(p, gq, bu, 0.01, 0.9, 1e-4, False, ema=em), p)
    rows.append(("sgd+ema fused", n, (), t, None, n * 28 / t / 1e6))
    t = timeit(ops, lambda: ops.ema(em, p, 0.99), p)
    rows.append(("ema", n, (), t, None, n * 12 / t / 1e6))
    print(f"{'kernel':28s} {'C':>8s} {'shape':>16s} {'ms':>9s} {'TFLOP/s':>9s} {'GB/s':>9s}")
    for name, c, sp, t, tf, gb in rows:
        print(f"{name:28s} {c:8d} {str(tuple(sp)):>16s} {t:9.4f} {'' if tf is None else f'{tf/1e3:9.2f}':>9s} {'' if gb is None else f'{gb:9.1f}':>9s}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_kernels.json"), "w") as f:
        json.dump([dict(kernel=r[0], C=r[1], shape=list(r[2]), ms=r[3], gflops=r[4], mbps=r[5]) for r in rows], f, indent=1)


if __name__ == "__main__":
    main()
