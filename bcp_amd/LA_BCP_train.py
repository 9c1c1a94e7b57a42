"""The LA train script of this build: the reference's command line (code/LA_BCP_train.py:32-54 -- every flag with its default)
and its two phases -- pre_train (:118-195) and self_train (:197-340) with the same iteration counts, LR schedule, validation
cadence and checkpoint files -- driving the fused step functions of bcp_amd/train_step.py (one grouped teacher forward, one
grouped student forward / backward per iteration).  Synthetic LA-like cases stand in for the h5 dataset
(dataloaders/dataset.py:SyntheticLA); TensorBoard snapshots (:294-340) are out of scope.

  python -m bcp_amd.LA_BCP_train --labelnum 8 --pre_max_iteration 20 --self_max_iteration 20
"""
import argparse
import logging
import os
import random
import sys

import numpy as np
import torch

from bcp_amd import plan, train_step
from bcp_amd.dataloaders.dataset import DeviceRotFlipCrop, SyntheticLA, TwoStreamBatchSampler, batches
from bcp_amd.networks.net_factory import net_factory
from bcp_amd.utils import ramps, test_3d_patch

# (flag, type, default) -- the reference's CLI, then this build's additions
_REFERENCE_FLAGS = (
    ("root_path", str, "/data/byh_data/SSNet_data/LA"), ("exp", str, "BCP"), ("model", str, "VNet"),
    ("pre_max_iteration", int, 2000), ("self_max_iteration", int, 15000), ("max_samples", int, 80),
    ("labeled_bs", int, 4), ("batch_size", int, 8), ("base_lr", float, 0.01), ("deterministic", int, 1),
    ("labelnum", int, 8), ("gpu", str, "1"), ("seed", int, 1337), ("consistency", float, 1.0),
    ("consistency_rampup", float, 40.0), ("magnitude", float, 10.0), ("u_weight", float, 0.5),
    ("mask_ratio", float, 2 / 3), ("u_alpha", float, 2.0), ("loss_weight", float, 0.5),
)
_BUILD_FLAGS = (
    ("fused_optimizer", int, 1, "1: the one-launch flat SGD; 0: torch.optim.SGD on the same parameters"),
    ("log_every", int, 50, "log (and synchronise with the host) every N iterations; the reference does so every iteration"),
    ("val_every", int, 200, "sliding-window validation cadence (the reference: every 200 iterations, :174,279)"),
    ("val_cases", int, 2, "synthetic validation volumes (the reference walks the LA test list)"),
)
parser = argparse.ArgumentParser()
for _name, _type, _default in _REFERENCE_FLAGS:
    parser.add_argument("--" + _name, type=_type, default=_default)
for _name, _type, _default, _help in _BUILD_FLAGS:
    parser.add_argument("--" + _name, type=_type, default=_default, help=_help)
parser.add_argument("--augment", action="store_true",
                    help="cases larger than the patch + the device-side RandomRotFlip / RandomCrop (dataloaders/dataset.py)")

patch_size = (112, 112, 80)
num_classes = 2


def save_net_opt(net, optimizer, path):
    """{'net', 'opt'} checkpoints, as the reference writes them (:79-84)"""
    torch.save({"net": net.state_dict(), "opt": optimizer.state_dict()}, str(path))


def load_net_opt(net, optimizer, path):
    ckpt = torch.load(str(path))
    net.load_state_dict(ckpt["net"])
    optimizer.load_state_dict(ckpt["opt"])


def load_net(net, path):
    net.load_state_dict(torch.load(str(path))["net"])


def get_current_consistency_weight(args, epoch):
    return args.consistency * ramps.sigmoid_rampup(epoch, args.consistency_rampup)


def _optimizer(args, model):
    if args.fused_optimizer:
        return train_step.FlatSGD(model, lr=args.base_lr, momentum=0.9, weight_decay=0.0001)
    return torch.optim.SGD(model.parameters(), lr=args.base_lr, momentum=0.9, weight_decay=0.0001)


def _data(args, device):
    """the labeled / unlabeled two-stream sampler over the training cases (:137-147): the LA h5 set under --root_path when its
    train.list exists (read once into a device-resident cache, dataloaders/h5_datasets.py), synthetic cases otherwise"""
    if os.path.exists(os.path.join(args.root_path, "train.list")):
        from bcp_amd.dataloaders.h5_datasets import LAHeart
        db = LAHeart(base_dir=args.root_path, split="train", num=args.max_samples, transform=DeviceRotFlipCrop(patch_size), device=device)
    else:
        logging.info("no {}/train.list: synthetic LA-like cases".format(args.root_path))
        db = SyntheticLA(num=args.max_samples, shape=patch_size, device=device, seed=args.seed,
                         transform=DeviceRotFlipCrop(patch_size) if args.augment else None,     # RandomRotFlip -> RandomCrop -> ToTensor (:122-126)
                         raw_shape=(patch_size[0] + 12, patch_size[1] + 10, patch_size[2] + 8))
    labeled, unlabeled = list(range(args.labelnum)), list(range(args.labelnum, args.max_samples))
    sampler = TwoStreamBatchSampler(labeled, unlabeled, args.batch_size, args.batch_size - args.labeled_bs)
    logging.info("{} iterations per epoch".format(len(sampler)))
    return db, sampler


def _val_cases(args, device):
    """stand-in for the LA test list (utils/test_3d_patch.py:22-25): volumes a little larger than the patch, so the sliding
    window takes several positions per axis"""
    from bcp_amd import synth
    shape = (patch_size[0] + 16, patch_size[1] + 8, patch_size[2] + 8)
    vols, labs = synth.la_batch(max(args.val_cases, 1), shape=shape, seed=args.seed + 99)
    return [(vols[i, 0].to(device), labs[i].to(device)) for i in range(args.val_cases)]


class _BestModel:
    """validation every `val_every` iterations; a better mean Dice writes iter_<n>_dice_<d>.pth and <model>_best_model.pth
    (:174-187 with the optimiser state, :279-293 weights only)"""

    def __init__(self, args, device, snapshot_path, with_optimizer):
        self.args, self.path, self.with_opt = args, snapshot_path, with_optimizer
        self.cases = _val_cases(args, device) if args.val_every > 0 else []
        self.best = 0
        self.validated = False

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
        dice = test_3d_patch.var_all_case_LA(model, num_classes=num_classes, patch_size=patch_size, stride_xy=18, stride_z=4, cases=self.cases)
        self.validated = True
        if dice > self.best:
            self.best = round(dice, 4)
            self._write(model, optimizer, "iter_{}_dice_{}.pth".format(iter_num, self.best))
            self._write(model, optimizer, "{}_best_model.pth".format(a.model))
            logging.info("save best model, dice %f" % self.best)

    def finish(self, model, optimizer):
        if self.best == 0:   # no checkpoint was written: keep the last weights so that the next phase can start
            if self.validated:
                logging.warning("validation ran but never scored above 0: %s_best_model.pth holds the LAST weights, not a best model", self.args.model)
            else:
                logging.info("no validation ran (--val_every 0): %s_best_model.pth holds the last weights", self.args.model)
            self._write(model, optimizer, "{}_best_model.pth".format(self.args.model))


def pre_train(args, snapshot_path, device):
    model = net_factory(net_type=args.model, in_chns=1, class_num=num_classes, mode="train")
    db_train, sampler = _data(args, device)
    optimizer = _optimizer(args, model)
    model.train()
    keeper = _BestModel(args, device, snapshot_path, with_optimizer=True)
    iter_num = 0
    while iter_num < args.pre_max_iteration:
        for sampled in batches(db_train, sampler):
            r = train_step.la_pre_train_step(model, optimizer, sampled["image"][:args.labeled_bs], sampled["label"][:args.labeled_bs], args.mask_ratio)
            iter_num += 1
            if iter_num % args.log_every == 0:
                logging.info("iteration %d : loss: %03f, loss_dice: %03f, loss_ce: %03f" % (iter_num, float(r["loss"]), float(r["loss_dice"]), float(r["loss_ce"])))
            keeper.maybe(iter_num, model, optimizer)
            if iter_num >= args.pre_max_iteration:
                break
    keeper.finish(model, optimizer)


def self_train(args, pre_snapshot_path, self_snapshot_path, device):
    model = net_factory(net_type=args.model, in_chns=1, class_num=num_classes, mode="train")
    ema_model = net_factory(net_type=args.model, in_chns=1, class_num=num_classes, mode="train")
    for p in ema_model.parameters():
        p.detach_()                     # the teacher never sees a gradient
    model.volatile_io = ema_model.volatile_io = True      # this loop consumes a pass's outputs before the network's next pass (networks/_hipnet.py)
    db_train, sampler = _data(args, device)
    optimizer = _optimizer(args, model)
    start = os.path.join(pre_snapshot_path, f"{args.model}_best_model.pth")
    load_net(model, start)              # student and teacher both start from the pre-trained weights (:220-222)
    load_net(ema_model, start)
    model.train()
    ema_model.train()
    keeper = _BestModel(args, device, self_snapshot_path, with_optimizer=False)
    iter_num = 0
    while iter_num < args.self_max_iteration:
        for sampled in batches(db_train, sampler):
            get_current_consistency_weight(args, iter_num // 150)   # computed and logged only, as in the reference (:246)
            r = train_step.la_self_train_step(model, ema_model, optimizer, sampled["image"], sampled["label"], args.labeled_bs,
                                              u_weight=args.u_weight, mask_ratio=args.mask_ratio, alpha=0.99)
            iter_num += 1
            if iter_num % args.log_every == 0:
                logging.info("iteration %d : loss: %03f, loss_l: %03f, loss_u: %03f" % (iter_num, float(r["loss"]), float(r["loss_l"]), float(r["loss_u"])))
            keeper.maybe(iter_num, model, optimizer)
            if iter_num % 2500 == 0:    # step decay (:273-276)
                for group in optimizer.param_groups:
                    group["lr"] = args.base_lr * 0.1 ** (iter_num // 2500)
            if iter_num >= args.self_max_iteration:
                break
    keeper.finish(model, optimizer)


def main(argv=None):
    args = parser.parse_args(argv)
    if args.deterministic:
        for seed_fn in (torch.manual_seed, random.seed, np.random.seed):
            seed_fn(args.seed)
    device = torch.device("cuda", torch.cuda.current_device())
    plan.use_real_stream(device)      # a real stream: what the capture of the recorded forward passes into HIP graphs needs (plan.GRAPHS = 1, the default; bcp_amd/plan.py)
    phase_dirs = ["./model/BCP/LA_{}_{}_labeled/{}".format(args.exp, args.labelnum, phase) for phase in ("pre_train", "self_train")]
    for d in phase_dirs:
        os.makedirs(d, exist_ok=True)
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s.%(msecs)03d] %(message)s", datefmt="%H:%M:%S", stream=sys.stdout)
    logging.info(str(args))
    pre_train(args, phase_dirs[0], device)
    self_train(args, phase_dirs[0], phase_dirs[1], device)


if __name__ == "__main__":
    main()
