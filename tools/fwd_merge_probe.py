"""What would merging the teacher and the student forward into one launch per layer buy?  Times (HIP events, no_grad, train-mode
BatchNorm): one net on 2 samples (solo), two nets on 2 samples each concurrently on two streams (what the step does), one net
on 4 samples in 4 groups (the launch shape a merged dual-network forward would have)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bcp_amd import synth
from bcp_amd.hip_ops import Ops
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); Ops.product()
model, ema = bench.build_models(dev, 1337)
vol, _ = synth.la_batch(4, seed=1); vol = vol.to(dev)
side = torch.cuda.Stream()
def t(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
with torch.no_grad():
    solo = t(lambda: model(vol[:2], groups=2))
    def both():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ema(vol[2:], groups=2)
        model(vol[:2], groups=2)
        main.wait_stream(side)
    conc = t(both)
    merged = t(lambda: model(vol, groups=4))
print(f"one net, 2 samples: {solo:.2f} ms | two nets concurrently (2 + 2 samples): {conc:.2f} ms | one net, 4 samples in 4 groups: {merged:.2f} ms")
