#!/usr/bin/env python
"""Golden vector for the validation path: the REFERENCE's utils/test_3d_patch.py:test_single_case driving the
REFERENCE V-Net in eval() mode on a small random volume.  Runs only in the build container (imports /root/reference through
oracle/make_golden.py's stubs); writes tests/golden/sw_la.npz (inputs + outputs: data, no reference source).

  python oracle/make_golden_eval.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
ARGV = list(sys.argv)        # make_golden resets sys.argv for the reference's argparse
import make_golden as MG  # noqa: E402  (installs the stubs, imports the reference)
import bcp_oracle as O  # noqa: E402

np.int = int  # the reference still spells `astype(np.int)` (test_3d_patch.py:135); numpy >= 1.24 dropped the alias
from utils import test_3d_patch as ref_t3d  # noqa: E402

SEED = 2024
PATCH, STRIDE_XY, STRIDE_Z = (32, 32, 16), 6, 4
SHAPE = (28, 40, 20)      # narrower than the patch along W: exercises the zero padding (:91-104)


def main():
    P = O.eval_params(SEED)
    net = MG.ref_vnet_la({k: v.clone() for k, v in P.items()})
    MG.set_drop_la(net, None)
    net.eval()
    rng = np.random.default_rng(SEED + 2)
    image = rng.standard_normal(SHAPE).astype(np.float32)
    label_map, score_map = ref_t3d.test_single_case(net, image, STRIDE_XY, STRIDE_Z, PATCH, num_classes=2)
    gt = (rng.random(SHAPE) < 0.4).astype(np.uint8)
    inter = int(((label_map != 0) & (gt != 0)).sum())
    dice = 2.0 * inter / (int((label_map != 0).sum()) + int(gt.sum()))
    out = os.path.join(HERE, "..", "tests", "golden", "sw_la.npz")
    np.savez_compressed(out, image=image, label_map=label_map.astype(np.uint8), score_map=score_map[0].astype(np.float32), gt=gt,
                        dice=np.float64(dice), seed=np.int64(SEED), patch=np.array(PATCH), stride=np.array([STRIDE_XY, STRIDE_Z]))
    print("wrote", os.path.normpath(out), "fg fraction", float(label_map.mean()), "dice vs random gt", dice,
          "score range", float(score_map.min()), float(score_map.max()))


def main_aug():
    """G10: the REFERENCE's RandomRotFlip + RandomCrop classes (dataloaders/dataset.py) on seeded volumes, np.random seeded per case"""
    from dataloaders import dataset as ref_ds
    rng = np.random.default_rng(SEED + 7)
    cases = []
    P = (24, 20, 16)
    for ci, shape in enumerate([(40, 36, 30), (30, 44, 16), (20, 22, 12), (24, 20, 16)]):   # plain, d == P (pad), all smaller (pad), equal (pad)
        image = rng.standard_normal(shape).astype(np.float32)
        label = (rng.random(shape) < 0.3).astype(np.uint8)
        for rep in range(3):
            seed = 100 * ci + rep
            np.random.seed(seed)
            out = ref_ds.RandomCrop(P)(ref_ds.RandomRotFlip()({"image": image, "label": label}))
            cases.append((ci, seed, out["image"].astype(np.float32), out["label"].astype(np.uint8)))
        cases_in = locals().setdefault("_ins", {})
        cases_in[ci] = (image, label)
    out = os.path.join(HERE, "..", "tests", "golden", "aug_la.npz")
    d = {"patch": np.array(P), "n_cases": np.int64(len(cases))}
    for ci, (image, label) in locals()["_ins"].items():
        d[f"in_image_{ci}"], d[f"in_label_{ci}"] = image, label
    for i, (ci, seed, img, lab) in enumerate(cases):
        d[f"case_{i}"] = np.array([ci, seed])
        d[f"out_image_{i}"], d[f"out_label_{i}"] = img, lab
    np.savez_compressed(out, **d)
    print("wrote", os.path.normpath(out), len(cases), "cases")


def main_aug_acdc():
    """G11: the REFERENCE's RandomGenerator (dataloaders/dataset.py:69-88: python random + np.random + scipy rotate / zoom) on seeded
    slices of ACDC-like shapes; seeds chosen so that all three branches (rot90 + flip, rotate, neither) occur"""
    import random
    from dataloaders import dataset as ref_ds
    rng = np.random.default_rng(SEED + 11)
    OUT = (64, 72)
    gen = ref_ds.RandomGenerator(OUT)
    d, n, branches = {"out_hw": np.array(OUT)}, 0, []
    for ci, shape in enumerate([(54, 64), (64, 54), (39, 47), (80, 91), (64, 72)]):
        image = rng.random(shape).astype(np.float32)                         # ACDC slices are min-max normalised floats
        label = (rng.random(shape) * 4).astype(np.uint8)
        # smooth-ish labels: blocks of 3 x 3 so that nearest-neighbour picks are visible as structure, not noise
        label = np.kron(label[::3, ::3], np.ones((3, 3), dtype=np.uint8))[:shape[0], :shape[1]]
        label = np.pad(label, ((0, shape[0] - label.shape[0]), (0, shape[1] - label.shape[1])))
        d[f"in_image_{ci}"], d[f"in_label_{ci}"] = image, label
        for rep in range(6):
            seed = 1000 * ci + rep
            random.seed(seed)
            np.random.seed(seed)
            r1 = random.random()
            br = "rotflip" if r1 > 0.5 else ("rotate" if random.random() > 0.5 else "none")
            branches.append(br)
            random.seed(seed)
            np.random.seed(seed)
            out = gen({"image": image, "label": label})
            d[f"case_{n}"] = np.array([ci, seed])
            d[f"out_image_{n}"], d[f"out_label_{n}"] = out["image"].numpy(), out["label"].numpy()
            n += 1
    d["n_cases"] = np.int64(n)
    out = os.path.join(HERE, "..", "tests", "golden", "aug_acdc.npz")
    np.savez_compressed(out, **d)
    print("wrote", os.path.normpath(out), n, "cases; branches:", {b: branches.count(b) for b in set(branches)})


def main_sw_pancreas():
    """G12: the REFERENCE's pancreas/test_util.py:test_single_case driving the REFERENCE IN-V-Net (pancreas/Vnet.py) in eval mode"""
    ref_tu = MG._load(os.path.join(MG.REF, "pancreas", "test_util.py"), "ref_pancreas_test_util")
    P = O.init_params(O.vnet_param_shapes(variant="pancreas"), seed=SEED + 20, random_affine=True)
    net = MG.ref_pvnet.VNet()
    net.load_state_dict(MG.clone_params(P), strict=True)
    net.eval()
    rng = np.random.default_rng(SEED + 21)
    shape, patch, sxy, sz = (36, 40, 30), (32, 32, 32), 6, 4      # D < patch: the zero-padding branch (:103-112)
    image = rng.standard_normal(shape).astype(np.float32)
    with torch.no_grad():
        label_map, score_map = ref_tu.test_single_case(net, image, sxy, sz, patch, num_classes=2)
    out = os.path.join(HERE, "..", "tests", "golden", "sw_pancreas.npz")
    np.savez_compressed(out, image=image, label_map=label_map.astype(np.uint8), score_map=score_map.astype(np.float32),
                        seed=np.int64(SEED + 20), patch=np.array(patch), stride=np.array([sxy, sz]))
    print("wrote", os.path.normpath(out), "fg fraction", float(label_map.mean()), "score range", float(score_map.min()), float(score_map.max()))


def main_sampler():
    """G13: the REFERENCE's TwoStreamBatchSampler (dataloaders/dataset.py:280-307) -- batches drawn under seeded np.random for the LA
    (8 labeled of 80, batch 4 / 2 labeled; 16 of 80, batch 8 / 4) and ACDC (136 of 1312, batch 24 / 12) configurations, two epochs each"""
    from dataloaders import dataset as ref_ds
    d = {}
    for ci, (n_lab, n_all, bs, sec_bs, seed) in enumerate([(8, 80, 4, 2, 1337), (16, 80, 8, 4, 7), (136, 1312, 24, 12, 1337), (5, 23, 5, 3, 3)]):
        np.random.seed(seed)
        sampler = ref_ds.TwoStreamBatchSampler(list(range(n_lab)), list(range(n_lab, n_all)), bs, sec_bs)
        epochs = [np.array([list(b) for b in sampler], dtype=np.int64) for _ in range(2)]
        d[f"cfg_{ci}"] = np.array([n_lab, n_all, bs, sec_bs, seed, len(sampler)])
        d[f"epoch0_{ci}"], d[f"epoch1_{ci}"] = epochs
    d["n"] = np.int64(4)
    out = os.path.join(HERE, "..", "tests", "golden", "sampler.npz")
    np.savez_compressed(out, **d)
    print("wrote", os.path.normpath(out), [d[f"epoch0_{i}"].shape for i in range(4)])


def main_aug_pancreas():
    """G14: the REFERENCE's pancreas RandomCrop / CenterCrop classes (pancreas/dataloaders.py:22-91) on seeded volumes -- larger than
    the patch, equal to it along one axis (<= triggers the padding of ALL axes), and smaller.  The module's h5py / torchvision
    imports are make_golden's empty stubs; the two classes exercised use numpy only."""
    ref_pdl = MG._load(os.path.join(MG.REF, "pancreas", "dataloaders.py"), "ref_pancreas_dataloaders")
    rng = np.random.default_rng(SEED + 14)
    P = (12, 10, 8)
    d, n = {"patch": np.array(P)}, 0
    for ci, shape in enumerate([(17, 14, 11), (12, 13, 10), (9, 12, 5), (13, 11, 9)]):
        image = rng.standard_normal(shape).astype(np.float32)
        label = (rng.random(shape) < 0.3).astype(np.uint8)
        d[f"in_image_{ci}"], d[f"in_label_{ci}"] = image, label
        for rep in range(3):
            seed = 500 * ci + rep
            np.random.seed(seed)
            oi, ol = ref_pdl.RandomCrop(P)([image, label.astype(np.float32)])
            d[f"case_{n}"] = np.array([ci, seed, 0])
            d[f"out_image_{n}"], d[f"out_label_{n}"] = oi.astype(np.float32), ol.astype(np.uint8)
            n += 1
        oi, ol = ref_pdl.CenterCrop(P)([image, label.astype(np.float32)])
        d[f"case_{n}"] = np.array([ci, 0, 1])
        d[f"out_image_{n}"], d[f"out_label_{n}"] = oi.astype(np.float32), ol.astype(np.uint8)
        n += 1
    d["n_cases"] = np.int64(n)
    out = os.path.join(HERE, "..", "tests", "golden", "aug_pancreas.npz")
    np.savez_compressed(out, **d)
    print("wrote", os.path.normpath(out), n, "cases")


def main_opt_layout():
    """the LAYOUT of the 'opt' entry of the reference's {'net','opt'} checkpoints (LA_BCP_train.py:79-84,
    ACDC_BCP_train.py:60-64: torch.optim.SGD over model.parameters(); pancreas: torch.optim.Adam, pancreas/dataloaders.py:182)
    after one real backward + step of the REFERENCE networks: which parameter indices carry state, the param_group fields, the
    state entry names.  Data only (indices / names / shapes), written as JSON."""
    import io
    import json
    out = {}
    P = O.eval_params(SEED)
    net = MG.ref_vnet_la({k: v.clone() for k, v in P.items()})
    MG.set_drop_la(net, None)
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=0.0001)     # LA_BCP_train.py:218
    x = torch.from_numpy(np.random.default_rng(SEED).standard_normal((2, 1, 16, 16, 16)).astype(np.float32))
    y = net(x)[0]
    y.square().mean().backward()
    opt.step()
    buf = io.BytesIO()
    torch.save({"net": net.state_dict(), "opt": opt.state_dict()}, buf)       # == save_net_opt's payload
    sd = torch.load(io.BytesIO(buf.getvalue()), weights_only=False)["opt"]
    params = list(net.parameters())
    out["la_sgd"] = {"n_params": len(params), "state_indices": sorted(int(k) for k in sd["state"]),
                     "state_entry_keys": sorted(next(iter(sd["state"].values())).keys()),
                     "param_group": {k: v for k, v in sd["param_groups"][0].items() if k != "params"},
                     "params_field": [int(sd["param_groups"][0]["params"][0]), int(sd["param_groups"][0]["params"][-1]), len(sd["param_groups"][0]["params"])],
                     "shapes_first5": [list(sd["state"][i]["momentum_buffer"].shape) for i in sorted(sd["state"])[:5]]}
    netu, _ = MG.ref_unet({k: v.clone() for k, v in O.init_params(O.unet_param_shapes(), seed=SEED).items()})
    optu = torch.optim.SGD(netu.parameters(), lr=0.01, momentum=0.9, weight_decay=0.0001)   # ACDC_BCP_train.py:223
    yu = netu(torch.from_numpy(np.random.default_rng(SEED + 1).random((2, 1, 32, 32)).astype(np.float32)))
    yu.square().mean().backward()
    optu.step()
    sdu = optu.state_dict()
    out["acdc_sgd"] = {"n_params": len(list(netu.parameters())), "state_indices": sorted(int(k) for k in sdu["state"])}
    netp = MG.ref_pvnet.VNet()                                                          # pancreas/dataloaders.py:11 (before DataParallel)
    netp.load_state_dict(O.init_params(O.vnet_param_shapes(variant="pancreas"), seed=SEED), strict=True)
    netp.train()
    if True:
        optp = torch.optim.Adam(netp.parameters(), lr=1e-3)
        yp = netp(torch.from_numpy(np.random.default_rng(SEED + 2).standard_normal((1, 1, 32, 32, 32)).astype(np.float32)))[0]
        yp.square().mean().backward()
        optp.step()
        sdp = optp.state_dict()
        out["pancreas_adam"] = {"n_params": len(list(netp.parameters())), "state_indices": sorted(int(k) for k in sdp["state"]),
                                "state_entry_keys": sorted(next(iter(sdp["state"].values())).keys()),
                                "param_group": {k: (list(v) if isinstance(v, tuple) else v) for k, v in sdp["param_groups"][0].items() if k != "params"}}
    path = os.path.join(HERE, "..", "tests", "golden", "opt_layout.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.normpath(path), {k: (v["n_params"], len(v["state_indices"])) for k, v in out.items()})


if __name__ == "__main__":
    if len(ARGV) > 1 and ARGV[1] == "opt":
        main_opt_layout()
        sys.exit(0)
    if len(ARGV) > 1 and ARGV[1] == "aug_pancreas":
        main_aug_pancreas()
        sys.exit(0)
    main()
    main_aug()
    main_aug_acdc()
    main_sw_pancreas()
    main_sampler()
    main_aug_pancreas()
    main_opt_layout()
