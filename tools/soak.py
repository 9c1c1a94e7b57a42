"""300 LA self-training steps back to back: loss stays finite, allocator footprint stays flat (side streams + record_stream)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bcp_amd import synth, train_step
from bcp_amd.hip_ops import Ops
from bcp_amd import plan
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); plan.use_real_stream(dev); Ops.product(); np.random.seed(1)   # real stream, as the training scripts
model, ema = bench.build_models(dev, 1337)
opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
vol, lab = synth.la_batch(4, seed=1337); vol, lab = vol.to(dev), lab.to(dev)
losses, mem = [], []
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for it in range(N):
    r = train_step.la_self_train_step(model, ema, opt, vol, lab, 2)
    if it % 50 == 49:
        torch.cuda.synchronize()
        losses.append(round(float(r["loss"]), 4)); mem.append(torch.cuda.memory_reserved() >> 20)
print("loss every 50 steps:", losses)
print("reserved MiB:", mem, "peak allocated MiB:", torch.cuda.max_memory_allocated() >> 20)
assert all(np.isfinite(losses)) and mem[-1] <= mem[1] * 1.05
sd = model.state_dict()
assert all(torch.isfinite(v.float()).all() for v in sd.values())
print("ok")
