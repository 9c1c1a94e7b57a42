"""measurement only: the largest-CC chain on pseudo-label-like inputs (run under rocprofv3 --kernel-trace --stats)"""
import sys, torch
sys.path.insert(0, "/root/repo")
from bcp_amd.hip_ops import Ops
ops = Ops.product(); dev = torch.device("cuda:0")
for kv in sys.argv[1:]:
    k, _, v = kv.partition("=")
    ops.set_option(k, v)
g = torch.Generator(device="cpu").manual_seed(0)
for p in (0.5, 0.1):
    seg = (torch.rand(2, 112, 112, 80, generator=g) < p).to(torch.uint8).to(dev)
    for _ in range(10):
        ops.cc_largest(seg, 1, 3)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for p in (0.5, 0.1):
    seg = (torch.rand(2, 112, 112, 80, generator=g) < p).to(torch.uint8).to(dev)
    e0.record()
    for _ in range(20):
        ops.cc_largest(seg, 1, 3)
    e1.record(); torch.cuda.synchronize()
    print(f"p={p}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per cc_largest")
# (round 6) from the logits: pseudo-label launch + chain against the one chain (bcp_plabel_cc_largest)
lg = torch.randn(2, 112, 112, 80, 2, generator=g).to(dev)
for name, fn in (("plabel_bin + cc_largest", lambda: ops.cc_largest(ops.plabel_bin(lg, 0.5), 1, 3)), ("plabel_cc_largest", lambda: ops.plabel_cc_largest(lg, 0.5, 3))):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
