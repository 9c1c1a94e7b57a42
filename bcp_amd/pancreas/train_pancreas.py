"""Counterpart of code/pancreas/train_pancreas.py (ema_cutmix :103-179, pretrain :50-101) on synthetic 96^3 data.
Module-level constants mirror :22-48.  Four streams (lab_a, lab_b, unlab_a, unlab_b) of `batch_size` samples
each, Adam(1e-3), u_weight left at mix_loss's default 0.5 exactly as the reference calls it (:160,164).

  python -m bcp_amd.pancreas.train_pancreas --pretraining_epochs 1 --self_training_epochs 1 --steps_per_epoch 3
"""
import argparse
import logging
import sys

import numpy as np
import torch

from bcp_amd import plan, synth, train_step
from bcp_amd.pancreas.Vnet import create_Vnet

seed_test = 2020
batch_size, lr = 2, 1e-3
pretraining_epochs, self_training_epochs = 60, 200
alpha = 0.99
label_percent = 20
connect_mode = 2
patch_size = 64


class _Streams(list):
    """the four loader streams (lab_a, lab_b, unlab_a, unlab_b) as views of ONE resident batch (.vols / .labs), so the
    grouped step can hand the teacher / student their two sub-batches without a concatenation"""


def _streams(device, n=4, bs=2, seed=seed_test):
    vols, labs = synth.la_batch(n * bs, shape=(96, 96, 96), seed=seed)
    vols, labs = vols.to(device), labs.to(device)
    st = _Streams((vols[i * bs:(i + 1) * bs], labs[i * bs:(i + 1) * bs]) for i in range(n))
    st.vols, st.labs, st.bs = vols, labs, bs
    return st


class _LoaderStreams:
    """--device_input_pipeline 1: the reference's four loaders (pancreas/dataloaders.py:185-195: lab_a, lab_b = the labeled list
    forwards / reversed with RandomCrop(96^3); unlab_a, unlab_b = the unlabeled list forwards / reversed with CenterCrop) over
    device-resident synthetic cases -- every call draws the next batch of each and crops it on the device."""

    def __init__(self, device, bs, n_cases=4, data_root=None, list_dir=None, split_name="pancreas", labelp=10):
        from bcp_amd.pancreas.dataloaders import Pancreas, SyntheticPancreas
        if data_root and list_dir:        # the reference's h5 cases, read once into a device-resident cache (:185-195)
            mk = lambda split, rev: Pancreas(data_root, split_name, split=split, labelp=labelp, reverse=rev, list_dir=list_dir, device=device)
            self.sets = [mk("train_lab", False), mk("train_lab", True), mk("train_unlab", False), mk("train_unlab", True)]
        else:
            self.sets = [SyntheticPancreas("train_lab", device, n_cases), SyntheticPancreas("train_lab", device, n_cases, reverse=True),
                         SyntheticPancreas("train_unlab", device, n_cases), SyntheticPancreas("train_unlab", device, n_cases, reverse=True)]
        self.bs, self.at = bs, 0

    def __call__(self):
        items = [[ds[(self.at * self.bs + i) % len(ds)] for i in range(self.bs)] for ds in self.sets]
        self.at += 1
        vols = torch.stack([it[0] for its in items for it in its])
        labs = torch.stack([it[1] for its in items for it in its]).long()
        bs = self.bs
        st = _Streams((vols[i * bs:(i + 1) * bs], labs[i * bs:(i + 1) * bs]) for i in range(4))
        st.vols, st.labs, st.bs = vols, labs, bs
        return st


def pretrain(net1, optimizer, streams, steps):
    """train_pancreas.py:50-101: copy-paste of the two labeled streams, supervised (CE + Dice) / 2"""
    net1.train()
    for _ in range(steps):
        st = streams() if callable(streams) else streams
        vols, labs = torch.cat([st[0][0], st[1][0]]), torch.cat([st[0][1], st[1][1]])
        r = train_step.la_pre_train_step(net1, optimizer, vols, labs, variant="pancreas")
    return r["loss"]


def ema_cutmix(net, ema_net, optimizer, streams, steps, dp=None, grouped=None):
    """train_pancreas.py:103-179 through train_step.la_self_train_step(variant='pancreas').  grouped (default: whenever the four
    streams are views of one resident batch): the two teacher calls and the two student calls of an iteration are launched as one
    grouped forward each -- InstanceNorm statistics are per sample, so this is the same arithmetic in half the launches;
    grouped=False issues the reference's four separate network calls."""
    net.train()
    ema_net.train()
    if grouped is None:
        grouped = True
    for _ in range(steps):
        st = streams() if callable(streams) else streams
        if isinstance(st, _Streams) and len(st) == 4:
            vols, labs, bs = st.vols, st.labs, st.bs
        else:
            vols, labs, bs = torch.cat([s_[0] for s_ in st]), torch.cat([s_[1] for s_ in st]), st[0][0].shape[0]
        r = train_step.la_self_train_step(net, ema_net, optimizer, vols, labs, 2 * bs, variant="pancreas", connect_mode=connect_mode,
                                          alpha=alpha, dp=dp, grouped=grouped)
    return r["loss"]


def _val_set(device, n, seed=seed_test + 77):
    """stand-in for the reference's test list (h5 files): synthetic volumes a little larger than the 96^3 patch"""
    vols, labs = synth.la_batch(max(n, 1), shape=(104, 100, 96), seed=seed)
    return [(vols[i, 0].to(device), labs[i].to(device)) for i in range(max(n, 1))]


def main(argv=None):
    from pathlib import Path
    from bcp_amd.pancreas.pancreas_utils import load_net_opt, save_net, save_net_opt
    from bcp_amd.pancreas.test_util import test_calculate_metric
    ap = argparse.ArgumentParser()
    ap.add_argument("--pretraining_epochs", type=int, default=pretraining_epochs)
    ap.add_argument("--self_training_epochs", type=int, default=self_training_epochs)
    ap.add_argument("--steps_per_epoch", type=int, default=10)
    ap.add_argument("--batch_size", type=int, default=batch_size)
    ap.add_argument("--val_every", type=int, default=20, help="validate every N epochs (the reference's pretrain_save_step = st_save_step = 20, train_pancreas.py:36-37: gates best_ema*_pre.pth and the restart from it); 0 = off")
    ap.add_argument("--val_cases", type=int, default=1)
    ap.add_argument("--val_stride", type=int, nargs=2, default=[18, 4], help="sliding-window strides (xy, z); test_calculate_metric's defaults")
    ap.add_argument("--result_dir", type=str, default="result/cutmix")
    ap.add_argument("--data_root", type=str, default="", help="directory of the pancreas h5 cases (the reference's data_root, train_pancreas.py:41); with --list_dir: train on them")
    ap.add_argument("--list_dir", type=str, default="", help="directory holding <split_name>/<10|20>percent/{train_lab,train_unlab,test}.txt (the reference hard-codes its own, pancreas/dataloaders.py:103-106)")
    ap.add_argument("--labelp", type=int, default=10)
    ap.add_argument("--device_input_pipeline", type=int, default=0, help="1: draw every batch from the four loader streams of the reference (RandomCrop / CenterCrop to 96^3, pancreas/dataloaders.py) with the crops done on the device")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, stream=sys.stdout)
    np.random.seed(seed_test)
    torch.manual_seed(seed_test)
    device = torch.device("cuda", torch.cuda.current_device())
    plan.use_real_stream(device)      # a real stream: what the capture of the recorded forward passes into HIP graphs needs (plan.GRAPHS = 1, the default)
    net, ema_net = create_Vnet(), create_Vnet(ema=True)
    net.volatile_io = ema_net.volatile_io = True      # the loops below consume a pass's outputs before the network's next pass (networks/_hipnet.py)
    ema_net.load_state_dict(net.state_dict())
    optimizer = train_step.FlatAdam(net, lr=lr)
    if args.data_root and args.list_dir:
        streams = _LoaderStreams(device, args.batch_size, data_root=args.data_root, list_dir=args.list_dir, labelp=args.labelp)
    else:
        streams = _LoaderStreams(device, args.batch_size) if args.device_input_pipeline else _streams(device, 4, args.batch_size)
    val = _val_set(device, args.val_cases) if args.val_every else None
    pre_dir, st_dir = Path(args.result_dir) / "pretrain", Path(args.result_dir) / "self_train"
    if val is not None:
        pre_dir.mkdir(parents=True, exist_ok=True)
        st_dir.mkdir(parents=True, exist_ok=True)

    def validate(model):
        avg, _ = test_calculate_metric(model, val, num_classes=2, dim=(96, 96, 96), s_xy=args.val_stride[0], s_z=args.val_stride[1])
        return float(avg[0])

    max_dice, best_pre = -1.0, pre_dir / f"best_ema{label_percent}_pre.pth"
    for ep in range(1, args.pretraining_epochs + 1):
        if val is not None and ep % args.val_every == 0:             # train_pancreas.py:66-79
            val_dice = validate(net)
            if val_dice > max_dice:
                save_net_opt(net, optimizer, best_pre, ep)
                max_dice = val_dice
            logging.info("Evaluation: val_dice: %.4f, val_maxdice: %.4f", val_dice, max_dice)
        loss = pretrain(net, optimizer, streams, args.steps_per_epoch)
        logging.info("pretrain epoch %d loss %f", ep, float(loss.detach()))
    if val is not None and best_pre.exists():                         # :115-117 -- both nets start from the best pre-trained state
        load_net_opt(net, optimizer, best_pre)
        load_net_opt(ema_net, optimizer, best_pre)
    else:
        ema_net.load_state_dict(net.state_dict())
    max_dice = -1.0
    for ep in range(1, args.self_training_epochs + 1):
        if val is not None and ep % args.val_every == 0:             # :126-141
            val_dice = validate(net)
            if val_dice > max_dice:
                save_net(net, st_dir / f"best_ema_{label_percent}_self.pth")
                max_dice = val_dice
            logging.info("Evaluation: val_dice: %.4f, val_maxdice: %.4f", val_dice, max_dice)
        loss = ema_cutmix(net, ema_net, optimizer, streams, args.steps_per_epoch)
        logging.info("self-train epoch %d loss %f", ep, float(loss.detach()))


if __name__ == "__main__":
    main()
