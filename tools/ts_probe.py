"""Where does a deep-level conv workgroup's life go?  Needs a library built with -DBCP_TS_DEBUG=1 (tools/_abl/ts.so): wave 0 of the first
64 workgroups of k_c3f stamps s_memtime at its phase boundaries.   python tools/ts_probe.py tools/_abl/ts.so [C] [sk]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bcp_amd import _lib  # noqa: E402
from bcp_amd.hip_ops import Ops  # noqa: E402

lib = sys.argv[1]
Cc = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ops = Ops(_lib.Binding(lib))
if len(sys.argv) > 3:
    ops.set_option("conv3_b6_flat_sk", int(sys.argv[3]))
dev = torch.device("cuda:0")
sp = {128: (14, 14, 10), 256: (7, 7, 5), 64: (28, 28, 20), 32: (56, 56, 40)}[Cc]
x = torch.randn(2, *sp, Cc, device=dev)
w = torch.randn(Cc, Cc, 3, 3, 3, device=dev) * 0.05
wf, wd = ops.conv3_pack(w, 3)
BW = os.environ.get("TS_BW") == "1"          # probe the dgrad + backward-statistics instance instead of the plain forward
if BW:
    from bcp_amd import hip_ops as H
    g1, b1, rm, rv = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    yprev = torch.randn(2, *sp, Cc, device=dev)
    stp = ops.norm_fwd(yprev, 2, g1, b1, rm, rv, H.ACT_RELU)[1]
    run = lambda: ops.conv3_dgrad_bwdstats(x, wd, Cc, 3, yprev, stp, H.ACT_RELU, 2)
else:
    run = lambda: ops.conv3_fwd(x, wd, None, Cc, 3)
for _ in range(3):
    y = run()
torch.cuda.synchronize()
ops.b.cdll.bcp_debug_ts_clear()
y = run()
torch.cuda.synchronize()
span = (C.c_ulonglong * (8192 * 2))()
fs = ops.b.cdll.bcp_debug_ts_span
fs.argtypes = [C.c_void_p]
assert fs(span) == 0
import numpy as np
sp_ = np.frombuffer(span, dtype=np.uint64).reshape(8192, 2).astype(np.int64)
sp_ = sp_[(sp_[:, 0] > 0) & (sp_[:, 1] > 0)]
if len(sp_):
    t0 = sp_[:, 0].min()
    st, en = (sp_[:, 0] - t0) / 100.0, (sp_[:, 1] - t0) / 100.0      # us
    print(f"C={Cc}: {len(sp_)} workgroups; launch span first start -> last end {en.max():.2f} us; lifetimes us: min {np.min(en - st):.2f} median {np.median(en - st):.2f} max {np.max(en - st):.2f}")
    order = np.argsort(st)
    hist, edges = np.histogram(st, bins=12)
    print("  starts (us) histogram:", " ".join(f"{edges[i]:.1f}:{hist[i]}" for i in range(len(hist))))
    hist, edges = np.histogram(en, bins=12)
    print("  ends   (us) histogram:", " ".join(f"{edges[i]:.1f}:{hist[i]}" for i in range(len(hist))))
    busy = float(np.sum(en - st))
    print(f"  sum of lifetimes {busy:.0f} us = {busy / en.max():.0f} workgroups resident on average (512 slots at two per CU)")
    late = st > 1.0
    if late.any():
        print(f"  second-wave workgroups (start > 1 us): {int(late.sum())}, lifetimes median {np.median((en - st)[late]):.2f} us; first-wave median {np.median((en - st)[~late]):.2f} us")
buf = (C.c_ulonglong * (64 * 64))()
fn = ops.b.cdll.bcp_debug_ts
fn.argtypes = [C.c_void_p]
assert fn(buf) == 0
import numpy as np
a = np.frombuffer(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
print(f"C={Cc}: s_memtime ticks (~ shader clocks) since the workgroup's own start (the counters of different XCDs have different bases)")
print("        setup  first-barrier | stage durations ... | epilogue | lifetime")
life = []
clk = []
for wg in range(64):
    r = a[wg]
    if r[0] == 0 or r[61] == 0:
        continue
    stages = [i for i in range(3, 59) if r[i] > 0]
    pts = [r[0], r[1], r[2]] + [r[i] for i in stages] + [r[60], r[61]]
    d = np.diff(np.array(pts))
    life.append((r[61] - r[0], d))
    if r[62] > r[59] > 0:
        clk.append((r[61] - r[0]) / ((r[62] - r[59]) * 10.0))      # ticks per ns: s_memtime ticks against the 100 MHz constant clock
    if wg in (0, 1, 7, 8, 31, 63):
        print(f"wg {wg:2d}: {d[0]:6d} {d[1]:6d} | " + " ".join(f"{v:5d}" for v in d[2:-2]) + f" | {d[-2]:6d} {d[-1]:6d} | {r[61] - r[0]:7d}")
rt = a[(a[:, 59] > 0) & (a[:, 62] > 0)]
if len(rt):
    print("constant-clock view of the recorded workgroups: starts spread over %.2f us, first start -> last end %.2f us" % ((rt[:, 59].max() - rt[:, 59].min()) / 100.0, (rt[:, 62].max() - rt[:, 59].min()) / 100.0))
if clk:
    print("s_memtime ticks per ns over the workgroup's life (= the clock it ran at, GHz): median %.3f  min %.3f  max %.3f" % (np.median(clk), min(clk), max(clk)))
if life:
    order = sorted(range(len(life)), key=lambda i: life[i][0])
    print('lifetimes (ticks) by recorded workgroup:', ' '.join(str(int(life[i][0])) for i in range(len(life))))
    for tag, i in (('fastest', order[0]), ('slowest', order[-1])):
        d = life[i][1]
        print(f'{tag} workgroup: setup {d[0]} first {d[1]} stages ' + ' '.join(str(int(v)) for v in d[2:-2]) + f' | {d[-2]} epilogue {d[-1]}')
    L = np.array([l for l, _ in life])
    print("workgroup lifetime (ticks): min %d median %d max %d over %d workgroups" % (L.min(), np.median(L), L.max(), len(L)))
    D = np.array([d for _, d in life if len(d) == len(life[0][1])])
    med = np.median(D, axis=0)
    print("median per phase: setup %d, first fetch + stash + barrier %d, stages total %d (median stage %d, %d stages), last stage -> epilogue start %d, epilogue %d"
          % (med[0], med[1], med[2:-2].sum(), np.median(med[2:-2]), len(med) - 4, med[-2], med[-1]))
