"""Reproducer (GPU): the 2-D upsample kernel bcp_bilinear2x_fwd alone, launched back to back on W worker streams (fixed inputs, a ring of
output buffers each) with a load generator on one more stream; every output is compared bit for bit with the result of the same launch
made on an otherwise idle GPU.  Run from the root of the tree whose library is to be tested (round-4 tree: tools/_abl/r04head).

  python tools/probe/bilinear_race_probe.py [rounds=40] [workers=2] [load=1] [C=32] [H=16]
"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from bcp_amd.hip_ops import Ops

kv = dict(a.split("=") for a in sys.argv[1:])
ROUNDS, W, LOAD, C, H = int(kv.get("rounds", 40)), int(kv.get("workers", 2)), int(kv.get("load", 1)), int(kv.get("C", 32)), int(kv.get("H", 16))
RING = int(kv.get("ring", 64))
GEMM = int(kv.get("gemm", 0))          # 1: every upsample launch is preceded by the 1x1-conv GEMM that writes its input (as in the U-Net's decoder)
CONV = int(kv.get("conv", 0))          # streams running the U-Net's 2-D convs (bf16-pipe kernels, LDS-DMA weight stages) beside the upsample
ops = Ops.product(); dev = torch.device("cuda:0")
Ops.AMAX = False
g = torch.Generator(device="cpu"); g.manual_seed(3)
zs = [torch.randn(4, 1, H, H, C, generator=g).to(dev) for _ in range(W)]
from bcp_amd import hip_ops as Hh
hs, bps, pwb = [], [], []
if GEMM:
    for w_ in range(W):
        h = torch.randn(4, 1, H, H, 2 * C, generator=g).to(dev)
        wt = (torch.randn(C, 2 * C, generator=g) * 0.1).to(dev).contiguous()
        hs.append(h); bps.append(ops.k2_pack(wt, 2 * C, C, Hh.PACK_PW_FWD)); pwb.append(torch.zeros(C, device=dev))
        ops.pw_fwd(h, bps[-1], pwb[-1], C, out=zs[w_])
    torch.cuda.synchronize()
gold = []
for z in zs:                                   # reference: the same launch on an idle GPU
    y = torch.zeros(4, 1, 2 * H, 2 * H, 2 * C, device=dev)
    ops.bilinear2x_fwd(z, y, C)
    torch.cuda.synchronize()
    gold.append(y.clone())
streams = [torch.cuda.Stream(device=dev) for _ in range(W)]
ls = torch.cuda.Stream(device=dev)
la = torch.randn(32 << 20, device=dev); lb = torch.empty_like(la); lm = torch.randn(2048, 2048, device=dev); lo = torch.empty_like(lm)
rings = [[torch.zeros_like(gold[0]) for _ in range(RING)] for _ in range(W)]
convs = []
for k in range(CONV):
    cs = torch.cuda.Stream(device=dev)
    items = []
    for (cc, hh) in ((16, 64), (32, 32), (64, 16), (128, 8)):
        x = torch.randn(4, 1, hh, hh, cc, device=dev)
        w = (torch.randn(cc, cc, 3, 3, device=dev) * 0.1).contiguous()
        wf, _ = ops.conv3_pack(w, 1)
        items.append((x, wf, torch.zeros(cc, device=dev), cc))
    convs.append((cs, items))
torch.cuda.synchronize()
bad = 0
for r in range(ROUNDS):
    for w in range(W):
        for y in rings[w]:
            y.zero_()
    torch.cuda.synchronize()
    if LOAD:
        with torch.cuda.stream(ls):
            for _ in range(8):
                lb.copy_(la); torch.mm(lm, lm, out=lo); la[: 1 << 20].add_(1.0)
    for i in range(RING):
        for cs, items in convs:
            with torch.cuda.stream(cs):
                x, wf, bias, cc = items[i % len(items)]
                ops.conv3_fwd(x, wf, bias, cc, 1)
        for w in range(W):
            with torch.cuda.stream(streams[w]):
                if GEMM:
                    ops.pw_fwd(hs[w], bps[w], pwb[w], C, out=zs[w])
                ops.bilinear2x_fwd(zs[w], rings[w][i], C)
    torch.cuda.synchronize()
    for w in range(W):
        for i, y in enumerate(rings[w]):
            if not torch.equal(y, gold[w]):
                bad += 1
                if bad <= 6:
                    d = (y != gold[w]).nonzero()
                    ch = sorted(set(int(v) for v in d[:, 4]))
                    px = sorted(set((int(a), int(b), int(c)) for a, b, c in zip(d[:, 0], d[:, 2], d[:, 3])))
                    print(f"round {r} worker {w} launch {i}: {d.shape[0]} elements differ; channels {ch[:20]}; pixels (n,h,w) {px[:4]}; got "
                          f"{[round(float(y[tuple(k)]), 5) for k in d[:4]]} want {[round(float(gold[w][tuple(k)]), 5) for k in d[:4]]}", flush=True)
print(f"RESULT bilinear probe rounds={ROUNDS} workers={W} gemm={GEMM} conv={CONV} load={LOAD} C={C} H={H}: {bad} of {ROUNDS * W * RING} launches gave other bits than the idle-GPU launch")
