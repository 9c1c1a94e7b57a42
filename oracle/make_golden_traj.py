#!/usr/bin/env python
"""Golden K = 5 step self-training trajectories ("K=5-step trajectory"), driven by the REFERENCE's own networks and
functions exactly as oracle/make_golden.py drives its 3- / 2-step ones -- plus the SAME trajectory in float64, so that the
fixture carries the reference's own fp32-vs-fp64 drift per step: the tolerance a second fp32 implementation can be held to.
Runs only in the build container (imports /root/reference through make_golden's stubs).  Writes tests/golden/la_traj5.npz and
tests/golden/acdc_traj5.npz (inputs are regenerated from seeds; data only).

  python oracle/make_golden_traj.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (installs the stubs, imports the reference)
import bcp_oracle as O  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")
STEPS = 5
ENSEMBLE = 8


def la(dtype, forced=None, shape=(32, 32, 16), jitter=None):
    """forced: per-step (plab_a, plab_b) to use INSTEAD of the run's own pseudo-labels (its own are still computed and returned).
    jitter: None, or a seed -- every floating-point parameter is moved by -1 / 0 / +1 ulp (x *= 1 + s * 2^-23) before the run: a
    stand-in for "another fp32 implementation of the same arithmetic", whose results differ from the reference's by roundings"""
    P0 = O.init_params(O.vnet_param_shapes(), seed=41, random_affine=True)
    if jitter is not None:
        rj = np.random.default_rng(9000 + jitter)
        for k, v in P0.items():
            if v.is_floating_point() and "running" not in k:
                sgn = torch.from_numpy(rj.integers(-1, 2, size=tuple(v.shape)).astype(np.float32))
                P0[k] = (v.double() * (1.0 + sgn.double() * 2.0 ** -23)).float()
    cast = (lambda P: {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in P.items()})
    model, ema = MG.ref_vnet_la(cast(P0)), MG.ref_vnet_la(cast(P0))
    if dtype == torch.float64:
        model.double(); ema.double()
    for p in ema.parameters():
        p.detach_()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=0.0001)
    rngd = np.random.default_rng(42)
    vol, lab = O.synth_la_batch(4, shape=shape, seed=77)
    vol = vol.to(dtype)
    traj, boxes, drops_all, plabs = [], [], [], []
    for it in range(STEPS):
        img_a, img_b, unimg_a, unimg_b = vol[:1], vol[1:2], vol[2:3], vol[3:4]
        lab_a, lab_b = lab[:1], lab[1:2]
        d = {k: MG.la_drop_masks(rngd, 1) for k in ("t_a", "t_b", "s_l", "s_u")}
        drops_all.append(d)
        dd = {k: {kk: vv.to(dtype) for kk, vv in v.items()} for k, v in d.items()}
        with torch.no_grad():
            MG.set_drop_la(ema, dd["t_a"]); ua, _ = ema(unimg_a)
            MG.set_drop_la(ema, dd["t_b"]); ub, _ = ema(unimg_b)
            plab_a = MG.ref_la.get_cut_mask(ua, nms=1)
            plab_b = MG.ref_la.get_cut_mask(ub, nms=1)
            own = (float(plab_a.sum()), float(plab_b.sum()))
            plabs.append((plab_a.clone(), plab_b.clone()))
            if forced is not None:
                plab_a, plab_b = forced[it][0].to(plab_a.dtype), forced[it][1].to(plab_b.dtype)
            bs = tuple(int(v * 2 / 3) for v in shape)
            w, h, z = (int(rngd.integers(0, shape[i] - bs[i])) for i in range(3))
            box = (w, h, z) + bs
            boxes.append(box)
            img_mask, loss_mask = O.box_to_mask(box, shape, 1)
        mixl_img = img_a * img_mask + unimg_a * (1 - img_mask)
        mixu_img = unimg_b * img_mask + img_b * (1 - img_mask)
        MG.set_drop_la(model, dd["s_l"]); outputs_l, _ = model(mixl_img)
        MG.set_drop_la(model, dd["s_u"]); outputs_u, _ = model(mixu_img)
        loss_l = MG.ref_bcp.mix_loss(outputs_l, lab_a, plab_a, loss_mask, u_weight=0.5)
        loss_u = MG.ref_bcp.mix_loss(outputs_u, plab_b, lab_b, loss_mask, u_weight=0.5, unlab=True)
        loss = loss_l + loss_u
        opt.zero_grad(); loss.backward(); opt.step()
        MG.ref_bcp.update_ema_variables(model, ema, 0.99)
        traj.append([loss.item(), loss_l.item(), loss_u.item(), own[0], own[1]])
    return np.array(traj), np.array(boxes), drops_all, plabs


def acdc(dtype):
    U0 = O.init_params(O.unet_param_shapes(), seed=51, random_affine=True)
    cast = (lambda P: {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in P.items()})
    model, mblocks = MG.ref_unet(cast(U0))
    ema, eblocks = MG.ref_unet(cast(U0))
    if dtype == torch.float64:
        model.double(); ema.double()
    for p in ema.parameters():
        p.detach_()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=0.0001)
    vol, lab = O.synth_acdc_batch(8, shape=(64, 64), seed=78)
    vol = vol.to(dtype)
    rngd = np.random.default_rng(52)
    traj, boxes, dropbits = [], [], []
    for it in range(STEPS):
        img_a, img_b, uimg_a, uimg_b = vol[:2], vol[2:4], vol[4:6], vol[6:8]
        lab_a, lab_b = lab[:2], lab[2:4]
        d = {k: MG.unet_drop_masks(rngd, 2, (64, 64)) for k in ("t_a", "t_b", "s_unl", "s_l")}
        dropbits.append([np.concatenate([np.packbits(d[k][f"d{i}"].numpy().astype(np.uint8)) for i in range(5)]) for k in ("t_a", "t_b", "s_unl", "s_l")])
        dd = {k: {kk: vv.to(dtype) for kk, vv in v.items()} for k, v in d.items()}
        with torch.no_grad():
            MG.set_drop_unet(eblocks, dd["t_a"]); pre_a = ema(uimg_a)
            MG.set_drop_unet(eblocks, dd["t_b"]); pre_b = ema(uimg_b)
            plab_a = MG.ref_acdc.get_ACDC_masks(pre_a, nms=1)
            plab_b = MG.ref_acdc.get_ACDC_masks(pre_b, nms=1)
            w, h = int(rngd.integers(0, 64 - 42)), int(rngd.integers(0, 64 - 42))
            box = (w, h, 42, 42)
            boxes.append(box)
            img_mask, loss_mask = O.box_to_mask(box, (64, 64), 2)
        MG.set_drop_unet(mblocks, dd["s_unl"]); out_unl = model(uimg_a * img_mask + img_a * (1 - img_mask))
        MG.set_drop_unet(mblocks, dd["s_l"]); out_l = model(img_b * img_mask + uimg_b * (1 - img_mask))
        unl_dice, unl_ce = MG.ref_acdc.mix_loss(out_unl, plab_a, lab_a, loss_mask, u_weight=0.5, unlab=True)
        l_dice, l_ce = MG.ref_acdc.mix_loss(out_l, lab_b, plab_b, loss_mask, u_weight=0.5)
        loss_ce, loss_dice = unl_ce + l_ce, unl_dice + l_dice
        loss = (loss_dice + loss_ce) / 2
        opt.zero_grad(); loss.backward(); opt.step()
        MG.ref_acdc.update_model_ema(model, ema, 0.99)
        traj.append([loss.item(), loss_dice.item(), loss_ce.item(), float(plab_a.sum()), float(plab_b.sum())])
    return np.array(traj), np.array(boxes), np.array(dropbits)


def la_fixture(name, shape):
    t32, boxes, drops, plabs = la(torch.float32, shape=shape)
    t64, boxes64, _, _ = la(torch.float64, shape=shape)
    assert np.array_equal(boxes, boxes64)
    # The free-running trajectory bifurcates: a random-init teacher's probabilities hover around the 0.5 threshold, rounding
    # flips pseudo-label voxels, the largest-CC filter then keeps different components, and the two runs optimise different
    # targets.  traj64f = the fp64 run FORCED onto the fp32 run's pseudo-labels: the drift that is left is arithmetic only, which
    # is what a second fp32 implementation (same forcing) can be held to.  plab_xor = voxels on which the forced fp64 run's OWN
    # pseudo-labels differ from the fp32 run's (the reference's own disagreement, voxel by voxel).
    t64f, _, _, plabs64 = la(torch.float64, forced=plabs, shape=shape)
    # ONE fp32 run is one sample of a chaotic process (the drift grows ~10x per step): ENSEMBLE = the reference's fp32 run repeated
    # with every parameter moved by <= 1 ulp, forced onto the same pseudo-labels -- the spread of "an fp32 implementation of this
    # arithmetic" around the fp64 trajectory.  drift_ens[member, step] = max |loss terms - traj64f|; member 0 = the unjittered run.
    ens = [np.abs(t32[:, :3] - t64f[:, :3]).max(1)]
    for j in range(ENSEMBLE):
        tj, _, _, _ = la(torch.float32, forced=plabs, shape=shape, jitter=j)
        ens.append(np.abs(tj[:, :3] - t64f[:, :3]).max(1))
    ens = np.array(ens)
    pb = np.array([[np.packbits(pa.numpy().astype(np.uint8).ravel()), np.packbits(pbb.numpy().astype(np.uint8).ravel())] for pa, pbb in plabs])
    xor = np.array([float((a32 != a64.to(a32.dtype)).sum() + (b32 != b64.to(b32.dtype)).sum()) for (a32, b32), (a64, b64) in zip(plabs, plabs64)])
    np.savez_compressed(os.path.join(OUT, name), traj=t32, traj64=t64, traj64f=t64f, drift_ens=ens, boxes=boxes, plab_bits=pb, plab_xor=xor,
                        drops=np.array([[np.concatenate([d[k]["x5"].numpy().ravel(), d[k]["x9"].numpy().ravel()]) for k in ("t_a", "t_b", "s_l", "s_u")]
                                        for d in drops]), param_seed=np.int64(41), data_seed=np.int64(77), shape=np.array(shape))
    print(name, "free   |loss32 - loss64| per step:", np.abs(t32[:, 0] - t64[:, 0]), "plab count diff:", np.abs(t32[:, 3:] - t64[:, 3:]).sum(1))
    print(name, "ensemble drift per step: max", ens.max(0), "median", np.median(ens, 0))
    print(name, "forced |loss32 - loss64f| per step:", np.abs(t32[:, :3] - t64f[:, :3]).max(1), "own plab xor:", xor, "plab sums", t32[:, 3:].sum(1))


def main():
    la_fixture("la_traj5.npz", (32, 32, 16))      # CPU (simulator) and GPU suites
    la_fixture("la_traj5m.npz", (64, 64, 32))     # GPU suite only: 32 values per channel at the deepest level instead of 4
    a32, aboxes, bits = acdc(torch.float32)
    a64, _, _ = acdc(torch.float64)
    np.savez_compressed(os.path.join(OUT, "acdc_traj5.npz"), traj=a32, traj64=a64, boxes=aboxes, dropbits=bits,
                        param_seed=np.int64(51), data_seed=np.int64(78), shape=np.array([64, 64]))
    print("ACDC |loss32 - loss64| per step:", np.abs(a32[:, 0] - a64[:, 0]), "plab diff:", np.abs(a32[:, 3:] - a64[:, 3:]).sum(1))


if __name__ == "__main__":
    main()
