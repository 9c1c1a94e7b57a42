"""Counterpart of the reference's code/LA_BCP_train.py: same CLI flags and defaults (:32-55), same loop
structure (pre_train :118-195, self_train :198-348), same callee names -- resolved against the HIP-backed
modules of this package.  Data: the LA h5 files are not part of the build, so the default dataset is the
synthetic stand-in (dataloaders/dataset.py:SyntheticLA); TensorBoard snapshots (:294-340) are out of scope.

  python -m bcp_amd.LA_BCP_train --labelnum 8 --batch_size 4 --labeled_bs 2 --pre_max_iteration 20 --self_max_iteration 40
"""
import argparse
import logging
import os
import random
import sys

import numpy as np
import torch

from bcp_amd import train_step
from bcp_amd.dataloaders.dataset import DeviceRotFlipCrop, SyntheticLA, TwoStreamBatchSampler, batches
from bcp_amd.networks.net_factory import net_factory
from bcp_amd.train_step import get_cut_mask
from bcp_amd.utils import ramps, test_3d_patch
from bcp_amd.utils.BCP_utils import context_mask, mix_loss, update_ema_variables
from bcp_amd.utils.losses import sup_loss_parts

parser = argparse.ArgumentParser()
parser.add_argument('--root_path', type=str, default='/data/byh_data/SSNet_data/LA', help='Name of Dataset')
parser.add_argument('--exp', type=str, default='BCP', help='exp_name')
parser.add_argument('--model', type=str, default='VNet', help='model_name')
parser.add_argument('--pre_max_iteration', type=int, default=2000, help='maximum pre-train iteration to train')
parser.add_argument('--self_max_iteration', type=int, default=15000, help='maximum self-train iteration to train')
parser.add_argument('--max_samples', type=int, default=80, help='maximum samples to train')
parser.add_argument('--labeled_bs', type=int, default=4, help='batch_size of labeled data per gpu')
parser.add_argument('--batch_size', type=int, default=8, help='batch_size per gpu')
parser.add_argument('--base_lr', type=float, default=0.01, help='maximum epoch number to train')
parser.add_argument('--deterministic', type=int, default=1, help='whether use deterministic training')
parser.add_argument('--labelnum', type=int, default=8, help='trained samples')
parser.add_argument('--gpu', type=str, default='1', help='GPU to use')
parser.add_argument('--seed', type=int, default=1337, help='random seed')
parser.add_argument('--consistency', type=float, default=1.0, help='consistency')
parser.add_argument('--consistency_rampup', type=float, default=40.0, help='consistency_rampup')
parser.add_argument('--magnitude', type=float, default='10.0', help='magnitude')
# -- setting of BCP
parser.add_argument('--u_weight', type=float, default=0.5, help='weight of unlabeled pixels')
parser.add_argument('--mask_ratio', type=float, default=2 / 3, help='ratio of mask/image')
# -- setting of mixup
parser.add_argument('--u_alpha', type=float, default=2.0, help='unlabeled image ratio of mixuped image')
parser.add_argument('--loss_weight', type=float, default=0.5, help='loss weight of unimage term')
# -- additions of this build
parser.add_argument('--fused_optimizer', type=int, default=1, help='1: one-launch FlatSGD; 0: torch.optim.SGD on the same parameters')
parser.add_argument('--log_every', type=int, default=50, help='host sync + log cadence (the reference syncs every iteration)')
parser.add_argument('--val_every', type=int, default=200, help='sliding-window validation cadence (LA_BCP_train.py:174,279: every 200 iterations)')
parser.add_argument('--augment', action='store_true', help='cases larger than the patch + the device-side RandomRotFlip / RandomCrop (dataloaders/dataset.py)')
parser.add_argument('--val_cases', type=int, default=2, help='synthetic validation volumes (the reference walks the LA test list)')

patch_size = (112, 112, 80)
num_classes = 2


def save_net_opt(net, optimizer, path):
    torch.save({'net': net.state_dict(), 'opt': optimizer.state_dict()}, str(path))


def load_net_opt(net, optimizer, path):
    state = torch.load(str(path))
    net.load_state_dict(state['net'])
    optimizer.load_state_dict(state['opt'])


def load_net(net, path):
    state = torch.load(str(path))
    net.load_state_dict(state['net'])


def get_current_consistency_weight(args, epoch):
    return args.consistency * ramps.sigmoid_rampup(epoch, args.consistency_rampup)


def _optimizer(args, model):
    if args.fused_optimizer:
        return train_step.FlatSGD(model, lr=args.base_lr, momentum=0.9, weight_decay=0.0001)
    return torch.optim.SGD(model.parameters(), lr=args.base_lr, momentum=0.9, weight_decay=0.0001)


def _val_cases(args, device):
    """stand-in for the LA test list (utils/test_3d_patch.py:22-25): volumes a little larger than the patch, so the sliding
    window takes several positions per axis"""
    from bcp_amd import synth
    shape = (patch_size[0] + 16, patch_size[1] + 8, patch_size[2] + 8)
    vols, labs = synth.la_batch(max(args.val_cases, 1), shape=shape, seed=args.seed + 99)
    return [(vols[i, 0].to(device), labs[i].to(device)) for i in range(args.val_cases)]


def pre_train(args, snapshot_path, device):
    model = net_factory(net_type=args.model, in_chns=1, class_num=num_classes, mode="train")
    db_train = SyntheticLA(num=args.max_samples, shape=patch_size, device=device, seed=args.seed,
                           transform=DeviceRotFlipCrop(patch_size) if args.augment else None,     # RandomRotFlip -> RandomCrop -> ToTensor (:122-126)
                           raw_shape=(patch_size[0] + 12, patch_size[1] + 10, patch_size[2] + 8))
    labeled_idxs = list(range(args.labelnum))
    unlabeled_idxs = list(range(args.labelnum, args.max_samples))
    batch_sampler = TwoStreamBatchSampler(labeled_idxs, unlabeled_idxs, args.batch_size, args.batch_size - args.labeled_bs)
    sub_bs = int(args.labeled_bs / 2)
    optimizer = _optimizer(args, model)
    model.train()
    logging.info("{} iterations per epoch".format(len(batch_sampler)))
    iter_num = 0
    best_dice = 0
    val_cases = _val_cases(args, device) if args.val_every > 0 else []
    max_epoch = args.pre_max_iteration // len(batch_sampler) + 1
    for epoch_num in range(max_epoch):
        for sampled_batch in batches(db_train, batch_sampler):
            volume_batch, label_batch = sampled_batch['image'][:args.labeled_bs], sampled_batch['label'][:args.labeled_bs]
            img_a, img_b = volume_batch[:sub_bs], volume_batch[sub_bs:]
            lab_a, lab_b = label_batch[:sub_bs], label_batch[sub_bs:]
            with torch.no_grad():
                img_mask, loss_mask = context_mask(img_a, args.mask_ratio)
            """Mix Input"""
            volume_batch = img_a * img_mask + img_b * (1 - img_mask)
            label_batch = lab_a * img_mask + lab_b * (1 - img_mask)
            outputs, _ = model(volume_batch)
            loss_ce, loss_dice = sup_loss_parts(outputs, label_batch)
            loss = (loss_ce + loss_dice) / 2
            iter_num += 1
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            if iter_num % args.log_every == 0:
                logging.info('iteration %d : loss: %03f, loss_dice: %03f, loss_ce: %03f' % (iter_num, float(loss.detach()), float(loss_dice.detach()), float(loss_ce.detach())))
            if args.val_every > 0 and iter_num % args.val_every == 0:       # LA_BCP_train.py:174-187
                model.eval()
                dice_sample = test_3d_patch.var_all_case_LA(model, num_classes=num_classes, patch_size=patch_size, stride_xy=18, stride_z=4,
                                                            cases=val_cases)
                if dice_sample > best_dice:
                    best_dice = round(dice_sample, 4)
                    save_net_opt(model, optimizer, os.path.join(snapshot_path, 'iter_{}_dice_{}.pth'.format(iter_num, best_dice)))
                    save_net_opt(model, optimizer, os.path.join(snapshot_path, '{}_best_model.pth'.format(args.model)))
                    logging.info("save best model, dice %f" % best_dice)
                model.train()
            if iter_num >= args.pre_max_iteration:
                break
        if iter_num >= args.pre_max_iteration:
            break
    if best_dice == 0:   # no validation ran (or none improved): keep the last weights so that self-training can start
        save_net_opt(model, optimizer, os.path.join(snapshot_path, '{}_best_model.pth'.format(args.model)))


def self_train(args, pre_snapshot_path, self_snapshot_path, device):
    model = net_factory(net_type=args.model, in_chns=1, class_num=num_classes, mode="train")
    ema_model = net_factory(net_type=args.model, in_chns=1, class_num=num_classes, mode="train")
    for param in ema_model.parameters():
        param.detach_()   # ema_model set
    db_train = SyntheticLA(num=args.max_samples, shape=patch_size, device=device, seed=args.seed,
                           transform=DeviceRotFlipCrop(patch_size) if args.augment else None,     # RandomRotFlip -> RandomCrop -> ToTensor (:122-126)
                           raw_shape=(patch_size[0] + 12, patch_size[1] + 10, patch_size[2] + 8))
    labeled_idxs = list(range(args.labelnum))
    unlabeled_idxs = list(range(args.labelnum, args.max_samples))
    batch_sampler = TwoStreamBatchSampler(labeled_idxs, unlabeled_idxs, args.batch_size, args.batch_size - args.labeled_bs)
    sub_bs = int(args.labeled_bs / 2)
    optimizer = _optimizer(args, model)
    pretrained_model = os.path.join(pre_snapshot_path, f'{args.model}_best_model.pth')
    load_net(model, pretrained_model)
    load_net(ema_model, pretrained_model)
    model.train()
    ema_model.train()
    logging.info("{} iterations per epoch".format(len(batch_sampler)))
    iter_num = 0
    best_dice = 0
    val_cases = _val_cases(args, device) if args.val_every > 0 else []
    max_epoch = args.self_max_iteration // len(batch_sampler) + 1
    lr_ = args.base_lr
    for epoch in range(max_epoch):
        for sampled_batch in batches(db_train, batch_sampler):
            volume_batch, label_batch = sampled_batch['image'], sampled_batch['label']
            img_a, img_b = volume_batch[:sub_bs], volume_batch[sub_bs:args.labeled_bs]
            lab_a, lab_b = label_batch[:sub_bs], label_batch[sub_bs:args.labeled_bs]
            unimg_a, unimg_b = volume_batch[args.labeled_bs:args.labeled_bs + sub_bs], volume_batch[args.labeled_bs + sub_bs:]
            with torch.no_grad():
                unoutput_a, _ = ema_model(unimg_a)
                unoutput_b, _ = ema_model(unimg_b)
                plab_a = get_cut_mask(unoutput_a, nms=1)
                plab_b = get_cut_mask(unoutput_b, nms=1)
                img_mask, loss_mask = context_mask(img_a, args.mask_ratio)
            consistency_weight = get_current_consistency_weight(args, iter_num // 150)  # logged only, as in the reference

            mixl_img = img_a * img_mask + unimg_a * (1 - img_mask)
            mixu_img = unimg_b * img_mask + img_b * (1 - img_mask)
            outputs_l, _ = model(mixl_img)
            outputs_u, _ = model(mixu_img)
            loss_l = mix_loss(outputs_l, lab_a, plab_a, loss_mask, u_weight=args.u_weight)
            loss_u = mix_loss(outputs_u, plab_b, lab_b, loss_mask, u_weight=args.u_weight, unlab=True)

            loss = loss_l + loss_u

            iter_num += 1
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            if iter_num % args.log_every == 0:
                logging.info('iteration %d : loss: %03f, loss_l: %03f, loss_u: %03f' % (iter_num, float(loss.detach()), float(loss_l.detach()), float(loss_u.detach())))

            update_ema_variables(model, ema_model, 0.99)

            if args.val_every > 0 and iter_num % args.val_every == 0:       # LA_BCP_train.py:279-293
                model.eval()
                dice_sample = test_3d_patch.var_all_case_LA(model, num_classes=num_classes, patch_size=patch_size, stride_xy=18, stride_z=4,
                                                            cases=val_cases)
                if dice_sample > best_dice:
                    best_dice = round(dice_sample, 4)
                    torch.save(model.state_dict(), os.path.join(self_snapshot_path, 'iter_{}_dice_{}.pth'.format(iter_num, best_dice)))
                    torch.save(model.state_dict(), os.path.join(self_snapshot_path, '{}_best_model.pth'.format(args.model)))
                    logging.info("save best model, dice %f" % best_dice)
                model.train()

            # change lr
            if iter_num % 2500 == 0:
                lr_ = args.base_lr * 0.1 ** (iter_num // 2500)
                for param_group in optimizer.param_groups:
                    param_group['lr'] = lr_
            if iter_num >= args.self_max_iteration:
                break
        if iter_num >= args.self_max_iteration:
            break
    if best_dice == 0:
        torch.save(model.state_dict(), os.path.join(self_snapshot_path, '{}_best_model.pth'.format(args.model)))


def main(argv=None):
    args = parser.parse_args(argv)
    if args.deterministic:
        torch.manual_seed(args.seed)
        random.seed(args.seed)
        np.random.seed(args.seed)
    device = torch.device("cuda", torch.cuda.current_device())
    pre_snapshot_path = "./model/BCP/LA_{}_{}_labeled/pre_train".format(args.exp, args.labelnum)
    self_snapshot_path = "./model/BCP/LA_{}_{}_labeled/self_train".format(args.exp, args.labelnum)
    print("Starting BCP training.")
    for snapshot_path in [pre_snapshot_path, self_snapshot_path]:
        os.makedirs(snapshot_path, exist_ok=True)
    logging.basicConfig(level=logging.INFO, format='[%(asctime)s.%(msecs)03d] %(message)s', datefmt='%H:%M:%S', stream=sys.stdout)
    logging.info(str(args))
    pre_train(args, pre_snapshot_path, device)
    self_train(args, pre_snapshot_path, self_snapshot_path, device)


if __name__ == "__main__":
    main()
