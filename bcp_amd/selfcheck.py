"""smoke(): one tiny LA self-training step on cuda:0 through libbcp_hip.so, checked against the oracle
(test infrastructure, imported here only as the checker -- see oracle/bcp_oracle.py header)."""
import numpy as np
import torch


def smoke():
    import bcp_oracle as O  # checker only
    from bcp_amd import synth, train_step
    from bcp_amd.hip_ops import Ops
    from bcp_amd.networks.VNet import VNet
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    dev = torch.device("cuda:0")
    Ops.product()  # raises if libbcp_hip.so is missing
    shape = (32, 32, 16)
    P = O.init_params(O.vnet_param_shapes(), seed=5, random_affine=True)
    nets = []
    for _ in range(2):
        n = VNet(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True).to(dev)
        n.load_state_dict({k: P[k].clone() for k in n.state_dict()})
        nets.append(n.flatten_().train())
    model, ema = nets
    for p in ema.parameters():
        p.detach_()
    vol, lab = synth.la_batch(4, shape=shape, seed=9)
    rng = np.random.default_rng(1)
    drops = {k: {"x5": torch.from_numpy((rng.random((1, 256)) < 0.5).astype(np.float32)),
                 "x9": torch.from_numpy((rng.random((1, 16)) < 0.5).astype(np.float32))} for k in ("t_a", "t_b", "s_l", "s_u")}
    box = (3, 5, 2, 21, 21, 10)
    opt = train_step.FlatSGD(model, lr=0.01)
    r = train_step.la_self_train_step(model, ema, opt, vol.to(dev), lab.to(dev), 2, box=box, drops=drops)
    torch.cuda.synchronize()
    Ps = {k: v.clone() for k, v in P.items()}
    Pt = {k: v.clone() for k, v in P.items()}
    ro = O.la_self_train_step(Ps, Pt, vol, lab, box, drops, 1)
    dl = abs(float(r["loss"]) - float(ro["loss"]))
    dpl = int((r["plab_a"].cpu().float() != ro["plab_a"]).sum() + (r["plab_b"].cpu().float() != ro["plab_b"]).sum())
    print(f"[smoke] HIP loss {float(r['loss']):.6f}  oracle {float(ro['loss']):.6f}  |d| {dl:.2e}  pseudo-label voxels differing {dpl}")
    assert dl < 1e-4, "smoke: loss differs from the oracle"
    assert dpl <= 4, "smoke: pseudo-labels differ from the oracle"
    # one more step must run (momentum path) and stay finite
    r2 = train_step.la_self_train_step(model, ema, opt, vol.to(dev), lab.to(dev), 2, box=box, drops=drops)
    assert bool(torch.isfinite(r2["loss"]))
    print("[smoke] ok")
