"""Isolated timings of the 3x3x3 conv kernels at the LA V-Net's in-step launch shapes (grouped batch of 2), A/B over
library options (bcp_set_option), interleaved rounds, HIP events on the launch stream.
Usage: python tools/bench_conv.py [--rounds 5] [--iters 20] [--json out.json] [--variants "name:opt=val,opt=val;..."]"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bcp_amd.hip_ops import Ops  # noqa: E402

LEVELS = [(16, (112, 112, 80)), (32, (56, 56, 40)), (64, (28, 28, 20)), (128, (14, 14, 10)), (256, (7, 7, 5))]


def timeit(ops, fn, like, iters):
    e0, e1 = ops.event(), ops.event()
    ops.event_record(e0, like)
    for _ in range(iters):
        fn()
    ops.event_record(e1, like)
    return ops.event_elapsed_ms(e0, e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--json", default=os.path.join(ROOT, "gpurun_out", "bench_conv.json"))
    ap.add_argument("--variants", default="fp32:conv3_b6=0,wgrad_b6=0;bf16pipe:conv3_b6=1,wgrad_b6=1")
    ap.add_argument("--ops", default="fwd_stats,dgrad,wgrad")
    ap.add_argument("--levels", default="16,32,64,128,256")
    ap.add_argument("--lib", default="", help="library variant to load instead of the product libbcp_hip.so (measurements)")
    a = ap.parse_args()
    variants = []
    for v in a.variants.split(";"):
        name, _, opts = v.partition(":")
        variants.append((name, [tuple(o.split("=")) for o in opts.split(",") if o]))
    if a.lib:
        from bcp_amd import _lib
        ops = Ops(_lib.Binding(a.lib))
    else:
        ops = Ops.product()
    dev = torch.device("cuda:0")
    N = a.batch
    want = set(a.ops.split(","))
    lv = set(int(x) for x in a.levels.split(","))
    res = {}
    for C, sp in LEVELS:
        if C not in lv:
            continue
        x = torch.randn(N, *sp, C, device=dev)
        dy = torch.randn(N, *sp, C, device=dev)
        # as in the step: the tensors carry their |max| -> the two-plane fp16 instances (option conv3_f16 = 0 in a variant switches them off)
        from bcp_amd import hip_ops as H
        x._bcp_amax = H.amax_slots(float(x.abs().max()), dev)
        dy._bcp_amax = H.amax_slots(float(dy.abs().max()), dev)
        w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
        b = torch.zeros(C, device=dev)
        wf, wd = ops.conv3_pack(w, 3)
        y = torch.empty(N, *sp, C, device=dev)
        dw = torch.empty_like(w)
        flops = 2.0 * N * sp[0] * sp[1] * sp[2] * 27 * C * C
        fns = {}
        if "fwd_stats" in want:
            fns["fwd_stats"] = lambda: ops.conv3_fwd_stats(x, wf, b, C, 3, N)
        if "dgrad" in want:
            fns["dgrad"] = lambda: ops.conv3_fwd(dy, wd, None, C, 3, out=y)
        if "wgrad" in want:
            fns["wgrad"] = lambda: ops.conv3_wgrad(x, dy, dw, 3)
        # round 3: the layer as the networks launch it (conv + norm forward; dgrad + norm backward) under the current options
        from bcp_amd import hip_ops as H
        g1, b1 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        vox_g = sp[0] * sp[1] * sp[2]
        st = [ops.norm_fwd(x, N, g1, b1, rm, rv, H.ACT_RELU)[1]]
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)

        def fwd_chain():
            if ops.norm_slabs_ok(N, vox_g, C):
                sk = ops.conv3_nslabs(x.shape, C, 3)
                if sk > 1:
                    return ops.norm_fwd_slabs(ops.conv3_fwd_raw(x, wf, C, 3, sk), sk, b, N, g1, b1, rm, rv, H.ACT_RELU)
            yy, part, nb = ops.conv3_fwd_stats(x, wf, b, C, 3, N)
            return ops.norm_fwd(yy, N, g1, b1, rm, rv, H.ACT_RELU, partial=part, nb=nb)

        def bwd_chain():
            if ops.norm_slabs_ok(N, vox_g, C):
                sk = ops.conv3_nslabs(dy.shape, C, 3)
                if sk > 1:
                    return ops.norm_bwd_slabs(x, ops.conv3_fwd_raw(dy, wd, C, 3, sk), sk, N, st[0], H.ACT_RELU, dg, db, True)
            da, part, nb = ops.conv3_dgrad_bwdstats(dy, wd, C, 3, x, st[0], H.ACT_RELU, N)
            return ops.norm_bwd(x, da, N, st[0], H.ACT_RELU, dg, db, True, partial=part, nb=nb)
        if C == 16 and ("c1_norm_fwd" in want or "c1_norm_bwd" in want):
            # the first layer (Cin = 1 -> 16) fused with its norm, at the level's spatial size (round 3)
            x1 = torch.randn(N, *sp, 1, device=dev)
            w1 = torch.randn(16, 1, 3, 3, 3, device=dev) * 0.1
            a1, st1 = ops.conv3_c1_norm_fwd(x1, w1, b, 3, N, g1, b1, rm, rv, H.ACT_RELU)
            if "c1_norm_fwd" in want:
                fns["c1_norm_fwd"] = lambda: ops.conv3_c1_norm_fwd(x1, w1, b, 3, N, g1, b1, rm, rv, H.ACT_RELU)
            if "c1_norm_bwd" in want:
                fns["c1_norm_bwd"] = lambda: ops.conv3_c1_norm_bwd(x1, w1, b, 3, N, st1, dy, H.ACT_RELU, dg, db, True)
        if "fwd_chain" in want:
            fns["fwd_chain"] = fwd_chain
        if "bwd_chain" in want:
            fns["bwd_chain"] = bwd_chain
        times = {(op, vn): [] for op in fns for vn, _ in variants}
        for r in range(a.rounds + 1):
            for vn, opts in variants:
                for k, v in opts:
                    ops.set_option(k, v)
                for op, fn in fns.items():
                    fn(); fn()
                    t = timeit(ops, fn, x, a.iters)
                    if r:
                        times[(op, vn)].append(t)
                for k, _ in opts:
                    ops.set_option(k)
        for (op, vn), ts in times.items():
            med = statistics.median(ts)
            res[f"{op}/C{C}/{vn}"] = {"us": med * 1e3, "tflops": flops / med / 1e9, "min_us": min(ts) * 1e3, "frac_157": flops / med / 1e9 / 157.3}
            print(f"{op:10s} C={C:3d} {vn:8s} {med * 1e3:8.1f} us  {flops / med / 1e9:7.1f} TFLOP/s  (min {min(ts) * 1e3:.1f})", flush=True)
    os.makedirs(os.path.dirname(a.json), exist_ok=True)
    with open(a.json, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
