"""The ACDC train script of this build: the reference's command line (code/ACDC_BCP_train.py:33-56 -- every flag with its
default) and its two phases -- pre_train (:193-302) and self_train (:304-443) with the same iteration counts, validation cadence
and checkpoint hand-off (the pre-trained {'net','opt'} restores the student's weights AND momentum, the teacher's weights) --
driving the fused step functions of bcp_amd/train_step.py on the HIP-backed 2-D U-Net.  Synthetic ACDC-like slices stand in for
the h5 dataset; per-volume validation is bcp_amd/utils/val_2d.py.

  python -m bcp_amd.ACDC_BCP_train --labelnum 7 --max_iterations 20 --pre_iterations 10
"""
import argparse
import logging
import os
import random
import sys

import numpy as np
import torch

from bcp_amd import plan, train_step
from bcp_amd.dataloaders.dataset import DeviceRandomGenerator, SyntheticACDC, TwoStreamBatchSampler, batches
from bcp_amd.networks.net_factory import BCP_net
from bcp_amd.utils import val_2d

# (flag, type, default) -- the reference's CLI, then this build's additions
_REFERENCE_FLAGS = (
    ("root_path", str, "/data/byh_data/SSNet_data/ACDC"), ("exp", str, "BCP"), ("model", str, "unet"),
    ("pre_iterations", int, 10000), ("max_iterations", int, 30000), ("batch_size", int, 24), ("deterministic", int, 1),
    ("base_lr", float, 0.01), ("patch_size", list, [256, 256]), ("seed", int, 1337), ("num_classes", int, 4),
    ("labeled_bs", int, 12), ("labelnum", int, 7), ("u_weight", float, 0.5), ("gpu", str, "0"),
    ("consistency", float, 0.1), ("consistency_rampup", float, 200.0), ("magnitude", float, 6.0), ("s_param", int, 6),
)
_BUILD_FLAGS = (
    ("log_every", int, 50, "log (and synchronise with the host) every N iterations"),
    ("val_every", int, 200, "validation cadence (the reference: every 200 iterations, :273,402)"),
    ("val_cases", int, 2, "synthetic validation volumes (the reference walks its val list)"),
)
parser = argparse.ArgumentParser()
for _name, _type, _default in _REFERENCE_FLAGS:
    parser.add_argument("--" + _name, type=_type, default=_default)
for _name, _type, _default, _help in _BUILD_FLAGS:
    parser.add_argument("--" + _name, type=_type, default=_default, help=_help)
parser.add_argument("--augment", action="store_true",
                    help="raw-size slices + the device-side RandomGenerator (rot90 / flip / rotate + zoom, dataloaders/dataset.py)")


def patients_to_slices(dataset, patiens_num):
    """labeled slices for a number of labeled patients (:181-191); "4" (BASELINE.json configs[0]) is not in the reference's
    table -> the first four patients' 84 slices"""
    return {"1": 32, "3": 68, "4": 84, "7": 136, "14": 256, "21": 396, "28": 512, "35": 664, "70": 1312}[str(patiens_num)]


def save_net_opt(net, optimizer, path):
    """{'net', 'opt'} checkpoints, as the reference writes them (:60-65)"""
    torch.save({"net": net.state_dict(), "opt": optimizer.state_dict()}, str(path))


def load_net(net, path):
    net.load_state_dict(torch.load(str(path))["net"])


def load_net_opt(net, optimizer, path):
    ckpt = torch.load(str(path))
    net.load_state_dict(ckpt["net"])
    optimizer.load_state_dict(ckpt["opt"])


def _loader(args, device):
    """BaseDataSets(split='train') over --root_path when its train_slices.list exists (h5 slices read once into a device-resident
    cache, dataloaders/h5_datasets.py), synthetic slices otherwise (:209-224)"""
    if os.path.exists(os.path.join(args.root_path, "train_slices.list")):
        from bcp_amd.dataloaders.h5_datasets import BaseDataSets
        db = BaseDataSets(base_dir=args.root_path, split="train", num=None, transform=DeviceRandomGenerator(args.patch_size), device=device)
    else:
        logging.info("no {}/train_slices.list: synthetic ACDC-like slices".format(args.root_path))
        db = SyntheticACDC(num=1312, shape=tuple(args.patch_size), device=device, seed=args.seed,
                           transform=DeviceRandomGenerator(args.patch_size) if args.augment else None,   # RandomGenerator (:209-211)
                           raw_shape=(216, 248))
    n_labeled = patients_to_slices(args.root_path, args.labelnum)
    sampler = TwoStreamBatchSampler(list(range(n_labeled)), list(range(n_labeled, len(db))), args.batch_size, args.batch_size - args.labeled_bs)
    return db, sampler


def _val_set(args, device):
    """stand-in for BaseDataSets(split='val'): volumes of 8 slices at the training resolution"""
    from bcp_amd import synth
    out = []
    for i in range(args.val_cases):
        vols, labs = synth.acdc_batch(8, shape=tuple(args.patch_size), seed=args.seed + 500 + i)
        out.append((vols[:, 0].unsqueeze(0).to(device), labs.unsqueeze(0).to(device)))       # [1,S,X,Y]
    return out


def _validate(model, val_set, num_classes):
    """per-class (dice, hd95) averaged over the validation volumes -> mean Dice (:274-283)"""
    total = 0.0
    for image, label in val_set:
        total = total + np.array(val_2d.test_single_volume(image, label, model, classes=num_classes), dtype=np.float64)
    return float(np.mean(total / max(len(val_set), 1), axis=0)[0])


class _BestModel:
    """validation every `val_every` iterations; a better mean Dice writes iter_<n>_dice_<d>.pth and <model>_best_model.pth
    (:273-295 with the optimiser state, :402-424 weights only)"""

    def __init__(self, args, device, snapshot_path, with_optimizer):
        self.args, self.path, self.with_opt = args, snapshot_path, with_optimizer
        self.val_set = _val_set(args, device) if args.val_every > 0 else []
        self.best = 0.0

    def _write(self, model, optimizer, name):
        target = os.path.join(self.path, name)
        if self.with_opt:
            save_net_opt(model, optimizer, target)
        else:
            torch.save(model.state_dict(), target)

    def maybe(self, iter_num, model, optimizer):
        a = self.args
        if a.val_every <= 0 or iter_num % a.val_every:
            return
        performance = _validate(model, self.val_set, a.num_classes)
        if performance > self.best:
            self.best = performance
            self._write(model, optimizer, "iter_{}_dice_{}.pth".format(iter_num, round(self.best, 4)))
            self._write(model, optimizer, "{}_best_model.pth".format(a.model))
        logging.info("iteration %d : mean_dice : %f" % (iter_num, performance))

    def finish(self, model, optimizer):
        if self.best == 0.0:
            self._write(model, optimizer, "{}_best_model.pth".format(self.args.model))


def _log(args, iter_num, r):
    if iter_num % args.log_every == 0:
        logging.info("iteration %d: loss: %f, mix_dice: %f, mix_ce: %f" % (iter_num, float(r["loss"]), float(r["loss_dice"]), float(r["loss_ce"])))


def pre_train(args, snapshot_path, device):
    model = BCP_net(in_chns=1, class_num=args.num_classes)
    db_train, sampler = _loader(args, device)
    optimizer = train_step.FlatSGD(model, lr=args.base_lr, momentum=0.9, weight_decay=0.0001)
    model.train()
    keeper = _BestModel(args, device, snapshot_path, with_optimizer=True)
    iter_num = 0
    while iter_num < args.pre_iterations:
        for sampled in batches(db_train, sampler):
            r = train_step.acdc_pre_train_step(model, optimizer, sampled["image"][:args.labeled_bs], sampled["label"][:args.labeled_bs])
            iter_num += 1
            _log(args, iter_num, r)
            keeper.maybe(iter_num, model, optimizer)
            if iter_num >= args.pre_iterations:
                break
    keeper.finish(model, optimizer)


def self_train(args, pre_snapshot_path, snapshot_path, device):
    model = BCP_net(in_chns=1, class_num=args.num_classes)
    ema_model = BCP_net(in_chns=1, class_num=args.num_classes, ema=True)
    model.volatile_io = ema_model.volatile_io = True      # this loop consumes a pass's outputs before the network's next pass (networks/_hipnet.py)
    db_train, sampler = _loader(args, device)
    optimizer = train_step.FlatSGD(model, lr=args.base_lr, momentum=0.9, weight_decay=0.0001)
    start = os.path.join(pre_snapshot_path, "{}_best_model.pth".format(args.model))
    load_net(ema_model, start)                   # teacher: weights; student: weights and momentum (:335-337)
    load_net_opt(model, optimizer, start)
    model.train()
    ema_model.train()
    keeper = _BestModel(args, device, snapshot_path, with_optimizer=False)
    iter_num = 0
    while iter_num < args.max_iterations:
        for sampled in batches(db_train, sampler):
            r = train_step.acdc_self_train_step(model, ema_model, optimizer, sampled["image"], sampled["label"], args.labeled_bs,
                                                u_weight=args.u_weight, alpha=0.99)
            iter_num += 1
            _log(args, iter_num, r)
            keeper.maybe(iter_num, model, optimizer)
            if iter_num >= args.max_iterations:
                break
    keeper.finish(model, optimizer)


def main(argv=None):
    args = parser.parse_args(argv)
    if args.deterministic:
        for seed_fn in (torch.manual_seed, random.seed, np.random.seed):
            seed_fn(args.seed)
    device = torch.device("cuda", torch.cuda.current_device())
    plan.use_real_stream(device)      # a real stream: what the capture of the recorded forward passes into HIP graphs needs (plan.GRAPHS = 1, the default; bcp_amd/plan.py)
    phase_dirs = ["./model/BCP/ACDC_{}_{}_labeled/{}".format(args.exp, args.labelnum, phase) for phase in ("pre_train", "self_train")]
    for d in phase_dirs:
        os.makedirs(d, exist_ok=True)
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s.%(msecs)03d] %(message)s", datefmt="%H:%M:%S", stream=sys.stdout)
    logging.info(str(args))
    pre_train(args, phase_dirs[0], device)
    self_train(args, phase_dirs[0], phase_dirs[1], device)


if __name__ == "__main__":
    main()
