"""measurement only: per-step GPU time of the first steps behind a device synchronisation (bench.py's timed region starts with one): an event behind
every step, the host running ahead as in the timed region.   python tools/probe/first_steps_probe.py [steps]"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from bcp_amd import plan
from bcp_amd.dp import DataParallel
from bcp_amd.hip_ops import Ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dp = DataParallel()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
plan.use_real_stream(dev)
Ops.product()
args = types.SimpleNamespace(workload="la", batch_size=4, labeled_bs=2)
step, info = bench.make_workload(args, dp, dev)
for _ in range(6):
    step()
for rep in range(3):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    print(f"rep {rep}: " + " ".join(f"{m:.3f}" for m in ms))
