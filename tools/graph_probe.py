"""Feasibility probe: capture one whole LA self-training step (3 streams, autograd, optimiser, EMA) in a HIP graph and
time replays against eager launches.  Fixed box / seeds (the probe ignores per-step randomness)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bcp_amd import synth, train_step  # noqa: E402
from bcp_amd.hip_ops import Ops  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
Ops.product()
np.random.seed(1337)
model, ema = bench.build_models(dev, 1337)
opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
vol, lab = synth.la_batch(4, seed=1337)
vol, lab = vol.to(dev), lab.to(dev)
box = (10, 12, 9, 74, 74, 53)


def step():
    return train_step.la_self_train_step(model, ema, opt, vol, lab, 2, box=box)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    r = step()
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms/step loss {float(r['loss']):.6f}", flush=True)

g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "thread_local"
with torch.cuda.graph(g, stream=s, capture_error_mode=mode):
    r = step()
torch.cuda.synchronize()
print("captured", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"graph replay: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step loss {float(r['loss']):.6f}", flush=True)
