"""Where the host (enqueue) time of the LA step goes: cProfile over N steps without waiting for the GPU inside the loop."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bcp_amd import synth, train_step
from bcp_amd.hip_ops import Ops
from bcp_amd import plan
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); plan.use_real_stream(dev); Ops.product(); np.random.seed(1337)
model, ema = bench.build_models(dev, 1337)
opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
vol, lab = synth.la_batch(4, seed=1337); vol, lab = vol.to(dev), lab.to(dev)
for _ in range(5): train_step.la_self_train_step(model, ema, opt, vol, lab, 2)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N): train_step.la_self_train_step(model, ema, opt, vol, lab, 2)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host {1e3*(t1-t0)/N:.2f} ms/step, gpu-inclusive {1e3*(t2-t0)/N:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(N):
    train_step.la_self_train_step(model, ema, opt, vol, lab, 2)
    pr.disable(); torch.cuda.synchronize(); pr.enable()       # empty queue before every step
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
