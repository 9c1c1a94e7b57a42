"""Segment timing of the LA step on the MAIN stream with HIP events (no profiler, so the host stays ahead of the GPU as in a
normal run): student forward, wait-for-teacher + loss, backward, optimiser + EMA.   python tools/step_segments.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from bcp_amd import synth, train_step
from bcp_amd.hip_ops import Ops
from bcp_amd.utils import BCP_utils as BU

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
ops = Ops.product(); np.random.seed(1337)
model, ema = bench.build_models(dev, 1337)
opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
vol, lab = synth.la_batch(4, seed=1337); vol, lab = vol.to(dev), lab.to(dev)
marks = []
orig_pair, orig_bwd, orig_step = BU.mix_loss_pair, torch.Tensor.backward, opt.step
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
def pair(*a, **k):
    marks.append(("fwd_done", ev())); r = orig_pair(*a, **k); marks.append(("loss_done", ev())); return r
BU.mix_loss_pair = pair
def step_():
    marks.append(("bwd_done", ev())); r = orig_step(); return r
opt.step = step_
for _ in range(3):
    train_step.la_self_train_step(model, ema, opt, vol, lab, 2)
torch.cuda.synchronize()
acc = {}
N = 20
for it in range(N):
    marks.clear()
    e0 = ev()
    train_step.la_self_train_step(model, ema, opt, vol, lab, 2)
    e1 = ev()
    torch.cuda.synchronize()
    t = {k: e0.elapsed_time(e) for k, e in marks}
    t["end"] = e0.elapsed_time(e1)
    for k, v in t.items(): acc[k] = acc.get(k, 0.0) + v / N
print({k: round(v, 3) for k, v in acc.items()})
print(f"student fwd {acc['fwd_done']:.2f} ms | join teacher + loss {acc['loss_done'] - acc['fwd_done']:.2f} | backward {acc['bwd_done'] - acc['loss_done']:.2f} | opt + ema {acc['end'] - acc['bwd_done']:.2f} | total {acc['end']:.2f}")
