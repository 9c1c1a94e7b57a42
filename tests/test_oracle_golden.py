"""The oracle (oracle/bcp_oracle.py) against the golden vectors captured from the imported
reference by oracle/make_golden.py.  CPU only.  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

import bcp_oracle as O

# thread count: tests/conftest.py caps torch at the cores this process may really use (cgroup quota / affinity)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.fixture(scope="module")
def meta(golden_dir):
    with open(os.path.join(golden_dir, "meta.json")) as f:
        return json.load(f)


def stats(t):
    a = t.detach().double().reshape(-1)
    return np.array([float(a.sum()), float(a.abs().sum()), float((a * a).sum().sqrt())])


def test_state_dict_keys(meta):
    for variant, key in (("la", "vnet_la_keys"), ("pancreas", "vnet_pancreas_keys")):
        shapes = O.vnet_param_shapes(variant=variant)
        assert [[k, list(v)] for k, v in shapes.items()] == meta[key]
    shapes = O.unet_param_shapes()
    assert [[k, list(v)] for k, v in shapes.items()] == meta["unet_keys"]
    assert len(meta["vnet_la_keys"]) == 259 and len(meta["unet_keys"]) == 226
    assert O.trainable_keys(O.vnet_param_shapes()) == meta["vnet_la_param_names"]
    assert O.trainable_keys(O.unet_param_shapes()) == meta["unet_param_names"]
    assert O.trainable_keys(O.vnet_param_shapes(variant="pancreas")) == meta["vnet_pancreas_param_names"]
    assert meta["bn_eps"] == 1e-5 and meta["bn_momentum"] == 0.1 and meta["in_eps"] == 1e-5
    assert meta["in_affine"] is False and meta["in_track"] is False


def test_vnet_la_tiny(golden_dir, meta):
    g = _load(golden_dir, "vnet_la_tiny.npz")
    P = O.init_params(O.vnet_param_shapes(), seed=meta["vnet_la_tiny"]["param_seed"], random_affine=True)
    keys = set(O.trainable_keys(P))
    Q = O._with_grad(P, keys)
    dm = {"x5": torch.from_numpy(g["drop_x5"]), "x9": torch.from_numpy(g["drop_x9"])}
    out = O.vnet_forward(Q, torch.from_numpy(g["x"]), dm, True, "la")
    loss = O.sup_loss_la(out, torch.from_numpy(g["tgt"]))
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy(), g["logits"], rtol=1e-5, atol=1e-6)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    names = [str(n) for n in g["grad_names"]]
    assert len(names) == meta["vnet_la_tiny"]["n_grads"]
    for n, st in zip(names, g["grad_stats"]):
        np.testing.assert_allclose(stats(Q[n].grad), st, rtol=2e-4, atol=1e-6, err_msg=n)
    np.testing.assert_allclose(Q["encoder.block_one.conv.0.weight"].grad.numpy(), g["grad_block_one_w"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(Q["decoder.block_five_up.conv.0.weight"].grad.numpy(), g["grad_five_up_w"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(P["encoder.block_one.conv.1.running_mean"].numpy(), g["rm_block_one"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(P["decoder.block_nine.conv.1.running_var"].numpy(), g["rv_block_nine"], rtol=1e-5, atol=1e-7)
    assert int(P["encoder.block_one.conv.1.num_batches_tracked"]) == int(g["nbt"]) == 1


@pytest.mark.slow
def test_vnet_la_full(golden_dir, meta):
    g = _load(golden_dir, "vnet_la_full.npz")
    m = meta["vnet_la_full"]
    P = O.init_params(O.vnet_param_shapes(), seed=m["param_seed"], random_affine=True)
    x, lab = O.synth_la_batch(1, seed=m["data_seed"])
    np.testing.assert_allclose(stats(x), g["x_stats"], rtol=1e-9)
    assert int(lab.sum()) == int(g["lab_sum"])
    keys = set(O.trainable_keys(P))
    Q = O._with_grad(P, keys)
    dm = {"x5": torch.from_numpy(g["drop_x5"]), "x9": torch.from_numpy(g["drop_x9"])}
    out = O.vnet_forward(Q, x, dm, True, "la")
    loss = O.sup_loss_la(out, lab)
    loss.backward()
    np.testing.assert_allclose(stats(out), g["logits_stats"], rtol=1e-5)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    for n, st in zip([str(n) for n in g["grad_names"]], g["grad_stats"]):
        np.testing.assert_allclose(stats(Q[n].grad)[1:], st[1:], rtol=5e-4, atol=1e-7, err_msg=n)


def test_boxes(meta):
    ops = meta["ops"]
    np.random.seed(1337)
    assert [list(O.box_la(np.random.randint)) for _ in range(4)] == ops["la_boxes_seed1337"]
    m, lm = O.box_to_mask(tuple(ops["la_boxes_seed1337"][-1]), (112, 112, 80), 2)
    assert int(m.sum()) == ops["la_mask_sum"] == 713292 and lm.shape == (2, 112, 112, 80)
    np.random.seed(1337)
    assert [list(O.box_acdc(np.random.randint)) for _ in range(4)] == ops["acdc_boxes_seed1337"]
    np.random.seed(2020)
    assert [list(O.box_pancreas(np.random.randint)) for _ in range(4)] == ops["pancreas_boxes_seed2020"]


def test_mixloss_la(golden_dir):
    g = _load(golden_dir, "mixloss_la.npz")
    a, b, mask = (torch.from_numpy(g[k]) for k in ("a", "b", "mask"))
    for key, kw in (("1", dict(u_weight=0.5)), ("2", dict(u_weight=0.5, unlab=True))):
        lo = torch.from_numpy(g["logits"]).requires_grad_(True)
        l = O.mix_loss_la(lo, a, b, mask, **kw)
        l.backward()
        assert abs(l.item() - float(g["l" + key])) < 1e-6
        np.testing.assert_allclose(lo.grad.numpy(), g["g" + key], rtol=1e-5, atol=1e-9)
    lo = torch.from_numpy(g["logits"]).requires_grad_(True)
    l = O.sup_loss_la(lo, a)
    l.backward()
    assert abs(l.item() - float(g["l3"])) < 1e-6
    np.testing.assert_allclose(lo.grad.numpy(), g["g3"], rtol=1e-5, atol=1e-9)
    assert abs(O.mix_loss_la(lo, a, b, mask, unlab=True).item() - float(g["lp"])) < 1e-6
    assert abs(O.mask_dice_loss(lo, a).item() - float(g["dice"])) < 1e-6
    # SURVEY 8c smoke values
    assert abs(float(g["l1"]) - 1.042518616) < 1e-6 and abs(float(g["l2"]) - 1.040563822) < 1e-6


def test_mixloss_acdc(golden_dir):
    g = _load(golden_dir, "mixloss_acdc.npz")
    a, b, mask = (torch.from_numpy(g[k]) for k in ("a", "b", "mask"))
    for key, kw in (("1", dict(u_weight=0.5, unlab=True)), ("2", dict(u_weight=0.5))):
        lo = torch.from_numpy(g["logits"]).requires_grad_(True)
        d, c = O.mix_loss_acdc(lo, a, b, mask, **kw)
        ((d + c) / 2).backward()
        assert abs(d.item() - float(g["d" + key])) < 1e-6 and abs(c.item() - float(g["c" + key])) < 1e-6
        np.testing.assert_allclose(lo.grad.numpy(), g["g" + key], rtol=1e-5, atol=1e-9)
    assert abs(float(g["d1"]) - 0.976003885) < 1e-6 and abs(float(g["c1"]) - 2.618426561) < 1e-6


def test_plabel_cc(golden_dir, meta):
    g = _load(golden_dir, "plabel_cc.npz")
    lo = torch.from_numpy(g["logits3d"])
    cut = O.get_cut_mask(lo)
    assert np.array_equal(cut.numpy().astype(np.uint8), g["cut"])
    assert cut[0, 0, 0, :4].tolist() == [1, 1, 1, 1]  # p == 0.5 exactly -> 1
    for conn, key in ((None, "cc26"), (2, "cc18"), (1, "cc6")):
        cc = O.largest_cc(cut, conn)
        assert cc.dtype == torch.float32
        assert np.array_equal(cc.numpy().astype(np.uint8), g[key]), key
    lo2 = torch.from_numpy(g["logits2d"])
    am = O.get_acdc_argmax(lo2)
    assert np.array_equal(am.numpy().astype(np.uint8), g["argmax"])
    assert np.array_equal(O.largest_cc_acdc(am).numpy().astype(np.uint8), g["argmax_cc"])
    assert meta["ops"]["cc_dtype"] == "torch.float32"


def test_ema(meta):
    ops = meta["ops"]
    shapes = O.vnet_param_shapes()
    A = O.init_params(shapes, seed=11, random_affine=True)
    B = O.init_params(shapes, seed=12, random_affine=True)
    O.ema_params(A, B, O.trainable_keys(shapes), 0.99)
    for k, st in ops["ema_la"].items():
        np.testing.assert_allclose(stats(B[k]), st, rtol=1e-6, atol=1e-9, err_msg=k)
    ushapes = O.unet_param_shapes()
    U0 = O.init_params(ushapes, seed=21, random_affine=True)
    U1 = O.init_params(ushapes, seed=22, random_affine=True)
    for k in U1:
        if k.endswith("running_mean"):
            U1[k] = U1[k] + 0.25
        if k.endswith("num_batches_tracked"):
            U0[k] = U0[k] + 250
            U1[k] = U1[k] + 3
    O.ema_state_dict(U0, U1, 0.99)
    for k, st in ops["ema_acdc"].items():
        np.testing.assert_allclose(stats(U1[k]), st, rtol=1e-6, atol=1e-9, err_msg=k)


def _unpack(bits, shape):
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(bits)[:n].reshape(shape).astype(np.float32))


def _unet_drops(g_or_bits, n, hw, prefix=None):
    dm = {}
    off = 0
    for i, c in enumerate(O.UNET_CH):
        shp = (n, c, hw[0] >> i, hw[1] >> i)
        if prefix is not None:
            dm[f"d{i}"] = _unpack(g_or_bits[f"{prefix}{i}"], shp)
        else:
            nb = (int(np.prod(shp)) + 7) // 8
            dm[f"d{i}"] = _unpack(g_or_bits[off:off + nb], shp)
            off += nb
    return dm


def test_unet_tiny(golden_dir, meta):
    g = _load(golden_dir, "unet_tiny.npz")
    P = O.init_params(O.unet_param_shapes(), seed=meta["unet_tiny"]["param_seed"], random_affine=True)
    keys = set(O.trainable_keys(P))
    Q = O._with_grad(P, keys)
    dm = _unet_drops(g, 2, (64, 64), prefix="drop_d")
    out = O.unet_forward(Q, torch.from_numpy(g["x"]), dm, True)
    np.testing.assert_allclose(out.detach().numpy(), g["logits"], rtol=1e-5, atol=1e-6)
    tgt = torch.from_numpy(g["tgt"])
    w, h, pw, ph = meta["unet_tiny"]["mask_box"]
    _, lm = O.box_to_mask((w, h, pw, ph), (64, 64), 2)
    d, c = O.mix_loss_acdc(out, tgt, (tgt + 1) % 4, lm, u_weight=0.5)
    loss = (d + c) / 2
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    for n, st in zip([str(n) for n in g["grad_names"]], g["grad_stats"]):
        np.testing.assert_allclose(stats(Q[n].grad), st, rtol=2e-4, atol=1e-6, err_msg=n)
    np.testing.assert_allclose(Q["decoder.up4.conv1x1.weight"].grad.numpy(), g["grad_up4_1x1_w"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(P["encoder.in_conv.conv_conv.1.running_var"].numpy(), g["rv_in"], rtol=1e-5)


def test_vnet_pancreas_tiny(golden_dir, meta):
    g = _load(golden_dir, "vnet_pancreas_tiny.npz")
    P = O.init_params(O.vnet_param_shapes(variant="pancreas"), seed=meta["vnet_pancreas_tiny"]["param_seed"])
    Q = O._with_grad(P, set(O.trainable_keys(P)))
    out = O.vnet_forward(Q, torch.from_numpy(g["x"]), None, True, "pancreas")
    np.testing.assert_allclose(out.detach().numpy(), g["logits"], rtol=1e-5, atol=1e-6)
    loss = O.sup_loss_la(out, torch.from_numpy(g["tgt"]))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    for n, st in zip([str(n) for n in g["grad_names"]], g["grad_stats"]):
        np.testing.assert_allclose(stats(Q[n].grad), st, rtol=2e-4, atol=1e-6, err_msg=n)


def test_la_trajectory(golden_dir, meta):
    """3 self-training steps (teacher fwd, pseudo-label+CC, mix, student fwd/bwd, SGD, EMA) with the
    oracle vs the trajectory the reference's own functions produced (LA_BCP_train.py:235-270)."""
    g = _load(golden_dir, "la_traj.npz")
    m = meta["la_traj"]
    shapes = O.vnet_param_shapes()
    Ps = O.init_params(shapes, seed=m["param_seed"], random_affine=True)
    Pt = {k: v.clone() for k, v in Ps.items()}
    tkeys = O.trainable_keys(shapes)
    vol, lab = O.synth_la_batch(4, shape=tuple(m["shape"]), seed=m["data_seed"])
    bufs = {}
    for it in range(m["steps"]):
        drops = {}
        for j, k in enumerate(("t_a", "t_b", "s_l", "s_u")):
            v = torch.from_numpy(g["drops"][it, j])
            drops[k] = {"x5": v[:256].view(1, 256), "x9": v[256:].view(1, 16)}
        r = O.la_self_train_step(Ps, Pt, vol, lab, tuple(int(v) for v in g["boxes"][it]), drops, 1)
        O.sgd_step(Ps, r["grads"], bufs, tkeys, lr=0.01)
        O.ema_params(Ps, Pt, tkeys, 0.99)
        ref = g["traj"][it]
        assert abs(r["loss"].item() - ref[0]) < 2e-5 and abs(r["loss_l"].item() - ref[1]) < 2e-5
        assert float(r["plab_a"].sum()) == ref[3] and float(r["plab_b"].sum()) == ref[4]
    for k, st in zip(meta["vnet_la_param_names"][:60], g["final_w_stats"]):
        np.testing.assert_allclose(stats(Ps[k]), st, rtol=1e-4, atol=1e-6, err_msg=k)
    for k, st in zip(meta["vnet_la_param_names"][:60], g["final_ema_stats"]):
        np.testing.assert_allclose(stats(Pt[k]), st, rtol=1e-4, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(Pt["decoder.block_nine.conv.1.running_mean"].numpy(), g["final_ema_rm"], rtol=1e-4, atol=1e-6)


def test_acdc_trajectory(golden_dir, meta):
    g = _load(golden_dir, "acdc_traj.npz")
    m = meta["acdc_traj"]
    shapes = O.unet_param_shapes()
    Ps = O.init_params(shapes, seed=m["param_seed"], random_affine=True)
    Pt = {k: v.clone() for k, v in Ps.items()}
    tkeys = O.trainable_keys(shapes)
    vol, lab = O.synth_acdc_batch(8, shape=tuple(m["shape"]), seed=m["data_seed"])
    bufs = {}
    for it in range(m["steps"]):
        drops = {k: _unet_drops(g["dropbits"][it, j], 2, tuple(m["shape"])) for j, k in enumerate(("t_a", "t_b", "s_unl", "s_l"))}
        r = O.acdc_self_train_step(Ps, Pt, vol, lab, tuple(int(v) for v in g["boxes"][it]), drops, 2, 2)
        O.sgd_step(Ps, r["grads"], bufs, tkeys, lr=0.01)
        O.ema_state_dict(Ps, Pt, 0.99)
        ref = g["traj"][it]
        assert abs(r["loss"].item() - ref[0]) < 2e-5 and abs(r["loss_dice"].item() - ref[1]) < 2e-5
        assert float(r["plab_a"].sum()) == ref[3] and float(r["plab_b"].sum()) == ref[4]
    for k, st in zip(meta["unet_param_names"][:40], g["final_w_stats"]):
        np.testing.assert_allclose(stats(Ps[k]), st, rtol=1e-4, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(Pt["encoder.in_conv.conv_conv.1.running_mean"].numpy(), g["final_ema_rm"], rtol=1e-4, atol=1e-6)


def test_sliding_window_oracle_vs_reference(golden_dir):
    """oracle restatement of utils/test_3d_patch.py:test_single_case (eval-mode V-Net) vs the reference's own output"""
    g = np.load(os.path.join(golden_dir, "sw_la.npz"))
    P = O.eval_params(int(g["seed"]))
    label, score = O.sliding_window_la(P, g["image"], int(g["stride"][0]), int(g["stride"][1]), tuple(int(v) for v in g["patch"]))
    assert np.abs(score - g["score_map"]).max() < 2e-6
    border = np.abs(g["score_map"] - 0.5) < 1e-5
    assert np.array_equal(label[~border].astype(np.uint8), g["label_map"][~border])
    assert abs(O.dice_binary(g["label_map"], g["gt"]) - float(g["dice"])) < 1e-12


def test_sliding_window_pancreas_oracle_vs_reference(golden_dir):
    """oracle restatement of pancreas/test_util.py:test_single_case (two channels, argmax) vs the reference's own output"""
    g = np.load(os.path.join(golden_dir, "sw_pancreas.npz"))
    P = O.init_params(O.vnet_param_shapes(variant="pancreas"), seed=int(g["seed"]), random_affine=True)
    label, score = O.sliding_window_pancreas(P, g["image"], int(g["stride"][0]), int(g["stride"][1]), tuple(int(v) for v in g["patch"]))
    assert score.shape == g["score_map"].shape and np.abs(score - g["score_map"]).max() < 2e-6
    border = np.abs(g["score_map"][1] - g["score_map"][0]) < 2e-5
    assert np.array_equal(label[~border].astype(np.uint8), g["label_map"][~border])


def test_input_pipelines_oracle_vs_reference(golden_dir):
    """8f-4: the oracle's restatements of the LA (RandomRotFlip + RandomCrop) and ACDC (RandomGenerator: rot90 / flip, scipy
    rotate + zoom at order 0) transforms == the reference's classes on the same random state (goldens made by
    oracle/make_golden_eval.py from the imported reference)"""
    import random
    g = np.load(os.path.join(golden_dir, "aug_la.npz"))
    P = tuple(int(v) for v in g["patch"])
    for i in range(int(g["n_cases"])):
        ci, seed = (int(v) for v in g[f"case_{i}"])
        np.random.seed(seed)
        oi, ol = O.la_rotflip_crop(g[f"in_image_{ci}"], g[f"in_label_{ci}"], P, lambda lo, hi: int(np.random.randint(lo, hi)))
        assert np.array_equal(oi, g[f"out_image_{i}"]) and np.array_equal(ol, g[f"out_label_{i}"]), i
    g = np.load(os.path.join(golden_dir, "aug_pancreas.npz"))          # pancreas RandomCrop / CenterCrop (pancreas/dataloaders.py:22-91)
    P = tuple(int(v) for v in g["patch"])
    for i in range(int(g["n_cases"])):
        ci, seed, center = (int(v) for v in g[f"case_{i}"])
        np.random.seed(seed)
        oi, ol = O.pancreas_crop([g[f"in_image_{ci}"], g[f"in_label_{ci}"]], P, None if center else (lambda lo, hi: int(np.random.randint(lo, hi))))
        assert np.array_equal(oi, g[f"out_image_{i}"]) and np.array_equal(ol, g[f"out_label_{i}"]), i
    g = np.load(os.path.join(golden_dir, "aug_acdc.npz"))
    out_hw = tuple(int(v) for v in g["out_hw"])
    for i in range(int(g["n_cases"])):
        ci, seed = (int(v) for v in g[f"case_{i}"])
        random.seed(seed)
        np.random.seed(seed)
        oi, ol = O.acdc_random_generator(g[f"in_image_{ci}"], g[f"in_label_{ci}"], out_hw, random.random,
                                         lambda lo, hi: int(np.random.randint(lo, hi)))
        assert np.array_equal(oi, g[f"out_image_{i}"]) and np.array_equal(ol, g[f"out_label_{i}"]), i


def test_two_stream_sampler_vs_reference(golden_dir):
    """A12: bcp_amd's TwoStreamBatchSampler == the reference's class, batch for batch, over two epochs on the same seeded np.random
    stream, for the LA / ACDC configurations and a ragged one (5 labeled of 23, batch 5 / 3: remainder dropped, unlabeled batches
    straddling permutations)"""
    from bcp_amd.dataloaders.dataset import TwoStreamBatchSampler
    g = np.load(os.path.join(golden_dir, "sampler.npz"))
    for ci in range(int(g["n"])):
        n_lab, n_all, bs, sec_bs, seed, length = (int(v) for v in g[f"cfg_{ci}"])
        np.random.seed(seed)
        sampler = TwoStreamBatchSampler(list(range(n_lab)), list(range(n_lab, n_all)), bs, sec_bs)
        assert len(sampler) == length
        for ep in range(2):
            got = np.array([list(b) for b in sampler], dtype=np.int64)
            assert np.array_equal(got, g[f"epoch{ep}_{ci}"]), (ci, ep)
