"""Finer anatomy of the forward phase of the LA step (HIP events on both streams, no profiler)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bcp_amd import synth, train_step
from bcp_amd.hip_ops import Ops
from bcp_amd.utils import BCP_utils as BU
from bcp_amd import plan
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); plan.use_real_stream(dev); Ops.product(); np.random.seed(1337)
model, ema = bench.build_models(dev, 1337)
opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
vol, lab = synth.la_batch(4, seed=1337); vol, lab = vol.to(dev), lab.to(dev)
marks = {}
def ev(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); globals()["marks"][name] = e
def wrap(obj, name, pre, post):
    orig = obj.forward
    def f(*a, **k):
        ev(pre); r = orig(*a, **k); ev(post); return r
    obj.forward = f
wrap(ema, "ema", "teacher_fwd_start", "teacher_fwd_end")
wrap(model, "model", "student_fwd_start", "student_fwd_end")
orig_cut = train_step.get_cut_mask
def cut(*a, **k):
    r = orig_cut(*a, **k); ev("teacher_cc_end"); return r
train_step.get_cut_mask = cut
orig_join = model._join_wgrad_stream
def join(like):
    ev("main_bwd_end"); r = orig_join(like); ev("wgrad_joined"); return r
model._join_wgrad_stream = join
orig_pair = BU.mix_loss_pair
def pair(*a, **k):
    ev("loss_start"); r = orig_pair(*a, **k); ev("loss_end"); return r
BU.mix_loss_pair = pair
for _ in range(3): train_step.la_self_train_step(model, ema, opt, vol, lab, 2)
torch.cuda.synchronize()
# run N steps back to back (the host runs ahead of the GPU as in bench.py); keep the events of every step, read them at the end
N = 20
allm = []
for it in range(N):
    marks = {}
    globals()["marks"] = marks
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    train_step.la_self_train_step(model, ema, opt, vol, lab, 2)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    allm.append((e0, dict(marks), e1))
torch.cuda.synchronize()
acc = {}
for e0, m, e1 in allm[5:]:
    for k, e in m.items(): acc[k] = acc.get(k, 0.0) + e0.elapsed_time(e) / (N - 5)
    acc["end"] = acc.get("end", 0.0) + e0.elapsed_time(e1) / (N - 5)
for k, v in sorted(acc.items(), key=lambda kv: kv[1]): print(f"{k:20s} {v:7.3f} ms")
