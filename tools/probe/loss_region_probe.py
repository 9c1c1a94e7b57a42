"""which torch ops launch the small kernels between the student's forward and backward (round 6)?  torch.profiler over one LA step, the
step's Python phases bracketed by record_function labels."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from bcp_amd import synth, train_step, plan
from bcp_amd.hip_ops import Ops
from bcp_amd.utils import BCP_utils as BU
from bcp_amd.networks import _hipnet
from torch.profiler import profile, ProfilerActivity, record_function
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); plan.use_real_stream(dev); Ops.product(); np.random.seed(1337)
model, ema = bench.build_models(dev, 1337)
model.volatile_io = ema.volatile_io = True
opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
vol, lab = synth.la_batch(4, seed=1337); vol, lab = vol.to(dev), lab.to(dev)


def label(obj, name, tag):
    orig = getattr(obj, name)
    def f(*a, **k):
        with record_function("PH:" + tag):
            return orig(*a, **k)
    setattr(obj, name, f)


label(BU, "mix_loss_pair", "mix_loss_pair")
label(train_step, "_backward", "_backward")
label(train_step, "get_cut_mask", "get_cut_mask")
label(BU, "mix", "mix")
label(BU, "update_ema_variables", "ema")
label(opt, "zero_grad", "zero_grad")
label(opt, "step", "opt.step")
label(model, "_run_backward", "net._run_backward")
label(model, "begin_backward", "net.begin_backward")
label(BU._MixLossPairTotalFn, "backward", "loss.backward") if False else None
orig_rb = model._run_backward
def rb(saved, dout):
    pl = model._plans_for().get(("bin", tuple(dout.shape)))
    print("dout is the plan's own input:", pl is not None and dout.data_ptr() == pl.static_in.data_ptr(), "contiguous", dout.is_contiguous())
    return orig_rb(saved, dout)
model._run_backward = rb
for _ in range(4): train_step.la_self_train_step(model, ema, opt, vol, lab, 2)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    train_step.la_self_train_step(model, ema, opt, vol, lab, 2)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
phases = [e for e in evs if e.name.startswith("PH:")]
for e in sorted(evs, key=lambda e: e.time_range.start):
    if e.name.startswith("aten::") and e.kernels:
        inside = [p.name[3:] for p in phases if p.time_range.start <= e.time_range.start and e.time_range.end <= p.time_range.end]
        print(f"{e.name:20s} {[k.name[:50] for k in e.kernels]} shapes? in {inside}")
