"""GPU diagnostic (round 3): per-tensor gradient error of the full-size ACDC step vs the fp32 oracle under different library
switches -- which tensors are off, and which switch moves them.   python tools/diag/acdc_grad_diag.py [batch labeled_bs]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bcp_oracle as O  # noqa: E402
import kernel_checks as K  # noqa: E402
import net_checks as NC  # noqa: E402
from bcp_amd import train_step  # noqa: E402
from bcp_amd.hip_ops import Ops  # noqa: E402
from bcp_amd.networks._hipnet import HipNet  # noqa: E402

torch.set_num_threads(16)
ops = Ops.product()
dev = torch.device("cuda:0")
batch, labeled_bs = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (24, 12)
hw, seed = (256, 256), 61
rng = np.random.default_rng(seed)
lsub, usub = labeled_bs // 2, (batch - labeled_bs) // 2
P = O.init_params(O.unet_param_shapes(), seed=seed + 1, random_affine=True)
vol, lab = O.synth_acdc_batch(batch, shape=hw, seed=seed + 2)
drops = {k: NC._rand_unet_drops(rng, lsub if k.startswith("s") else usub, hw) for k in ("t_a", "t_b", "s_unl", "s_l")}
box = (37, 61, int(hw[0] * 2 / 3), int(hw[1] * 2 / 3))
ro = O.acdc_self_train_step({k: v.clone() for k, v in P.items()}, {k: v.clone() for k, v in P.items()}, vol, lab, box, drops, lsub, usub)
ro64 = None
if os.environ.get("DIAG_FP64") == "1":
    P64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in P.items()}
    d64 = {k: {kk: vv.double() for kk, vv in v.items()} for k, v in drops.items()}
    ro64 = O.acdc_self_train_step({k: v.clone() for k, v in P64.items()}, {k: v.clone() for k, v in P64.items()}, vol.double(), lab, box, d64, lsub, usub)
plabs = (ro["plab_a"].to(torch.uint8).to(dev), ro["plab_b"].to(torch.uint8).to(dev))


def run(tag, opts=(), overlap=True, grouped=True):
    for k, v in opts:
        ops.set_option(k, v)
    HipNet.overlap_wgrad = overlap
    try:
        model, ema = NC.make_unet(P, dev, ops), NC.make_unet(P, dev, ops)
        for p in ema.parameters():
            p.detach_()
        r = train_step.acdc_self_train_step(model, ema, None, vol.to(dev), lab.to(dev), labeled_bs, box=box, drops=drops, plabs=plabs, grouped=grouped)
        torch.cuda.synchronize()
        params = dict(model.named_parameters())
        errs = sorted(((K.rel_l2(params[k].grad, g), k) for k, g in ro["grads"].items() if not NC.is_prenorm_bias(k, params) and float(g.norm()) > 1e-9), reverse=True)
        print(f"== {tag}: dloss {abs(float(r['loss']) - float(ro['loss'])):.2e}  median {errs[len(errs) // 2][0]:.2e}  tensors > 1e-2: {sum(e > 1e-2 for e, _ in errs)}")
        for e, k in errs[:12]:
            extra = ""
            if ro64 is not None:
                extra = f"   hip-vs-fp64 {K.rel_l2(params[k].grad, ro64['grads'][k]):.2e}  oracle32-vs-fp64 {K.rel_l2(ro['grads'][k], ro64['grads'][k]):.2e}"
            print(f"   {e:.3e}  {k}{extra}")
    finally:
        for k, _ in opts:
            ops.set_option(k)
        HipNet.overlap_wgrad = True


run("default")
run("all round-3 switches off", (("conv3_xcd", 0), ("fuse_bwd_stats", 0), ("norm_small", 0)))
run("no wgrad side stream", overlap=False)
run("ungrouped (four network calls)", grouped=False)
run("fp32-MFMA convs (conv3_b6 = 0, wgrad_b6 = 0)", (("conv3_b6", 0), ("wgrad_b6", 0)))
