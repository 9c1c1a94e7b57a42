"""fuse_head False vs True after 1..3 self-training steps (tests/net_checks.check_fused_head's setting), with the head's backward fused
through the norm or not: largest weight difference per setting and step count"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, torch
import net_checks as NC
import bcp_oracle as O
from bcp_amd import _lib, train_step
from bcp_amd.hip_ops import Ops
from bcp_amd.utils import BCP_utils as BU
dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
ops = Ops.product() if dev.type == "cuda" else Ops(_lib.Binding(os.path.join("tests", "_emu", "libbcp_emu.so")), allow_cpu=True)
if dev.type != "cuda":
    BU.set_test_ops(ops)
what = sys.argv[1] if len(sys.argv) > 1 else "la"
maxsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2

def run(fuse, headbwd, steps):
    type(ops).HEAD_BWD_FUSED = headbwd
    torch.manual_seed(5); np.random.seed(5)
    shape = (32, 32, 16) if what == "la" else (32, 32, 32)
    P = O.init_params(O.vnet_param_shapes(variant=what), seed=43, random_affine=True)
    model, ema = NC.make_vnet(P, dev, ops, what), NC.make_vnet(P, dev, ops, what)
    model.fuse_head = ema.fuse_head = fuse
    vol, lab = O.synth_la_batch(4, shape=shape, seed=79)
    model.seed_dropout(11); ema.seed_dropout(12)
    for p in ema.parameters():
        p.detach_()
    vol, lab = vol.to(dev), lab.to(dev)
    opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
    out = []
    for _ in range(steps):
        r = train_step.la_self_train_step(model, ema, opt, vol, lab, 2, box=(3, 5, 2, 21, 21, 10), variant=what,
                                          connect_mode=2 if what != "la" else None, grouped=True)
        out.append((float(r["loss"]), {k: v.detach().clone().cpu() for k, v in model.state_dict().items()}))
    return out

ref = run(False, False, maxsteps)
for headbwd in (False, True):
    got = run(True, headbwd, maxsteps)
    for s in range(maxsteps):
        worst = max(((float((ref[s][1][k] - got[s][1][k]).abs().max()) / max(float(ref[s][1][k].abs().max()), 1e-3), k) for k in ref[s][1] if ref[s][1][k].dtype.is_floating_point))
        print(f"{what} head_bwd_fused={headbwd} after step {s + 1}: loss {ref[s][0]:.7f} vs {got[s][0]:.7f}; worst relative weight difference {worst[0]:.3e} ({worst[1]})", flush=True)

if dev.type == "cuda":       # the two paths at the LA size, one call each: dy, dgamma / dbeta, dw / db, |max|
    import bcp_amd.hip_ops as H
    rng = np.random.default_rng(3)
    for (N, G, sp, drop) in ((2, 1, (112, 112, 80), True), (4, 2, (96, 96, 96), False)):
        y = (torch.randn(N, *sp, 16, device=dev) * 1.7 + 0.3)
        gamma, beta = torch.rand(16, device=dev) + 0.5, torch.randn(16, device=dev) * 0.2
        cs = (torch.randint(0, 2, (N, 16), device=dev).float() * 2) if drop else None
        w = (torch.randn(2, 16, 1, 1, 1, device=dev) * 0.3).contiguous()
        dlog = torch.randn(N, *sp, 2, device=dev) * 1e-3
        _, st = ops.norm_fwd(y, G, gamma, beta, torch.zeros(16, device=dev), torch.ones(16, device=dev), H.ACT_RELU, chan_scale=cs, stats_only=True)
        dw0, db0, dg0, dbe0 = torch.zeros_like(w), torch.zeros(2, device=dev), torch.zeros(16, device=dev), torch.zeros(16, device=dev)
        dw1, db1, dg1, dbe1 = torch.zeros_like(w), torch.zeros(2, device=dev), torch.zeros(16, device=dev), torch.zeros(16, device=dev)
        da = ops.pw16_bwd_norm(y, st, cs, G, H.ACT_RELU, dlog, w, dw0, db0, accumulate=True)
        ref = ops.norm_bwd(y, da, G, st, H.ACT_RELU, dg0, dbe0, True, chan_scale=cs)
        got = ops.pw16_bwd_norm_bwd(y, st, cs, G, H.ACT_RELU, dlog, w, dw1, db1, dg1, dbe1, norm_accumulate=True, accumulate=True)
        torch.cuda.synchronize()
        rel = lambda a, b: float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        print(f"full size N={N} G={G} {sp} drop={drop}: dy {rel(got, ref):.3e} (elements differing {int((got != ref).sum())} of {got.numel()}), dgamma {rel(dg1, dg0):.3e}, "
              f"dbeta {rel(dbe1, dbe0):.3e}, dw {rel(dw1, dw0):.3e}, db {rel(db1, db0):.3e}, |max| {H.amax_value(ops._amax_of(got)):.6e} vs {H.amax_value(ops._amax_of(ref)):.6e} "
              f"(tensor {float(got.abs().max()):.6e})", flush=True)
