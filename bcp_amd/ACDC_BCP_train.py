"""Counterpart of the reference's code/ACDC_BCP_train.py: same flags/defaults (:33-56), pre_train (:193-302) and
self_train (:304-443) loop structure and callee names, on the HIP-backed U-Net and kernels; synthetic ACDC-like
slices stand in for the h5 dataset; validation (val_2d) is a "next" row.

  python -m bcp_amd.ACDC_BCP_train --labelnum 7 --max_iterations 20 --pre_iterations 10
"""
import argparse
import logging
import os
import random
import sys

import numpy as np
import torch

from bcp_amd import train_step
from bcp_amd.dataloaders.dataset import DeviceRandomGenerator, SyntheticACDC, TwoStreamBatchSampler, batches
from bcp_amd.networks.net_factory import BCP_net
from bcp_amd.train_step import acdc_mix_loss as mix_loss, generate_mask, get_ACDC_masks, update_model_ema
from bcp_amd.utils import val_2d

parser = argparse.ArgumentParser()
parser.add_argument('--root_path', type=str, default='/data/byh_data/SSNet_data/ACDC', help='Name of Experiment')
parser.add_argument('--exp', type=str, default='BCP', help='experiment_name')
parser.add_argument('--model', type=str, default='unet', help='model_name')
parser.add_argument('--pre_iterations', type=int, default=10000, help='maximum epoch number to train')
parser.add_argument('--max_iterations', type=int, default=30000, help='maximum epoch number to train')
parser.add_argument('--batch_size', type=int, default=24, help='batch_size per gpu')
parser.add_argument('--deterministic', type=int, default=1, help='whether use deterministic training')
parser.add_argument('--base_lr', type=float, default=0.01, help='segmentation network learning rate')
parser.add_argument('--patch_size', type=list, default=[256, 256], help='patch size of network input')
parser.add_argument('--seed', type=int, default=1337, help='random seed')
parser.add_argument('--num_classes', type=int, default=4, help='output channel of network')
# label and unlabel
parser.add_argument('--labeled_bs', type=int, default=12, help='labeled_batch_size per gpu')
parser.add_argument('--labelnum', type=int, default=7, help='labeled data')
parser.add_argument('--u_weight', type=float, default=0.5, help='weight of unlabeled pixels')
# costs
parser.add_argument('--gpu', type=str, default='0', help='GPU to use')
parser.add_argument('--consistency', type=float, default=0.1, help='consistency')
parser.add_argument('--consistency_rampup', type=float, default=200.0, help='consistency_rampup')
parser.add_argument('--magnitude', type=float, default='6.0', help='magnitude')
parser.add_argument('--s_param', type=int, default=6, help='multinum of random masks')
parser.add_argument('--log_every', type=int, default=50)
parser.add_argument('--val_every', type=int, default=200, help='validation cadence (ACDC_BCP_train.py:273,402: every 200 iterations)')
parser.add_argument('--augment', action='store_true', help='raw-size slices + the device-side RandomGenerator (rot90 / flip / rotate + zoom, dataloaders/dataset.py)')
parser.add_argument('--val_cases', type=int, default=2, help='synthetic validation volumes (the reference walks its val list)')


def patients_to_slices(dataset, patiens_num):
    """:181-191; "4" (BASELINE.json configs[0]) is not in the reference's table -> 84 slices"""
    ref_dict = {"1": 32, "3": 68, "4": 84, "7": 136, "14": 256, "21": 396, "28": 512, "35": 664, "70": 1312}
    return ref_dict[str(patiens_num)]


def save_net_opt(net, optimizer, path):
    torch.save({'net': net.state_dict(), 'opt': optimizer.state_dict()}, str(path))


def load_net(net, path):
    net.load_state_dict(torch.load(str(path))['net'])


def load_net_opt(net, optimizer, path):
    state = torch.load(str(path))
    net.load_state_dict(state['net'])
    optimizer.load_state_dict(state['opt'])


def _loader(args, device):
    db_train = SyntheticACDC(num=1312, shape=tuple(args.patch_size), device=device, seed=args.seed,
                             transform=DeviceRandomGenerator(args.patch_size) if args.augment else None,   # RandomGenerator (:209-211)
                             raw_shape=(216, 248))
    labeled_slice = patients_to_slices(args.root_path, args.labelnum)
    labeled_idxs = list(range(0, labeled_slice))
    unlabeled_idxs = list(range(labeled_slice, len(db_train)))
    return db_train, TwoStreamBatchSampler(labeled_idxs, unlabeled_idxs, args.batch_size, args.batch_size - args.labeled_bs)


def _val_set(args, device):
    """stand-in for BaseDataSets(split='val'): volumes of 8 slices at the training resolution"""
    from bcp_amd import synth
    out = []
    for i in range(args.val_cases):
        vols, labs = synth.acdc_batch(8, shape=tuple(args.patch_size), seed=args.seed + 500 + i)
        out.append((vols[:, 0].unsqueeze(0).to(device), labs.unsqueeze(0).to(device)))       # [1,S,X,Y]
    return out


def _validate(model, val_set, num_classes):
    """ACDC_BCP_train.py:274-283: per-class (dice, hd95) averaged over the validation volumes -> mean Dice"""
    metric_list = 0.0
    for image, label in val_set:
        metric_list = metric_list + np.array(val_2d.test_single_volume(image, label, model, classes=num_classes), dtype=np.float64)
    metric_list = metric_list / max(len(val_set), 1)
    return float(np.mean(metric_list, axis=0)[0])


def pre_train(args, snapshot_path, device):
    labeled_sub_bs = int(args.labeled_bs / 2)
    model = BCP_net(in_chns=1, class_num=args.num_classes)
    db_train, batch_sampler = _loader(args, device)
    optimizer = train_step.FlatSGD(model, lr=args.base_lr, momentum=0.9, weight_decay=0.0001)
    model.train()
    iter_num = 0
    best_performance = 0.0
    val_set = _val_set(args, device) if args.val_every > 0 else []
    while iter_num < args.pre_iterations:
        for sampled_batch in batches(db_train, batch_sampler):
            volume_batch, label_batch = sampled_batch['image'], sampled_batch['label']
            img_a, img_b = volume_batch[:labeled_sub_bs], volume_batch[labeled_sub_bs:args.labeled_bs]
            lab_a, lab_b = label_batch[:labeled_sub_bs], label_batch[labeled_sub_bs:args.labeled_bs]
            img_mask, loss_mask = generate_mask(img_a)
            # -- original
            net_input = img_a * img_mask + img_b * (1 - img_mask)
            out_mixl = model(net_input)
            loss_dice, loss_ce = mix_loss(out_mixl, lab_a, lab_b, loss_mask, u_weight=1.0, unlab=True)
            loss = (loss_dice + loss_ce) / 2
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            iter_num += 1
            if iter_num % args.log_every == 0:
                logging.info('iteration %d: loss: %f, mix_dice: %f, mix_ce: %f' % (iter_num, float(loss.detach()), float(loss_dice.detach()), float(loss_ce.detach())))
            if args.val_every > 0 and iter_num % args.val_every == 0:            # ACDC_BCP_train.py:273-295
                performance = _validate(model, val_set, args.num_classes)
                if performance > best_performance:
                    best_performance = performance
                    save_net_opt(model, optimizer, os.path.join(snapshot_path, 'iter_{}_dice_{}.pth'.format(iter_num, round(best_performance, 4))))
                    save_net_opt(model, optimizer, os.path.join(snapshot_path, '{}_best_model.pth'.format(args.model)))
                logging.info('iteration %d : mean_dice : %f' % (iter_num, performance))
            if iter_num >= args.pre_iterations:
                break
    if best_performance == 0.0:
        save_net_opt(model, optimizer, os.path.join(snapshot_path, '{}_best_model.pth'.format(args.model)))


def self_train(args, pre_snapshot_path, snapshot_path, device):
    labeled_sub_bs, unlabeled_sub_bs = int(args.labeled_bs / 2), int((args.batch_size - args.labeled_bs) / 2)
    model = BCP_net(in_chns=1, class_num=args.num_classes)
    ema_model = BCP_net(in_chns=1, class_num=args.num_classes, ema=True)
    db_train, batch_sampler = _loader(args, device)
    optimizer = train_step.FlatSGD(model, lr=args.base_lr, momentum=0.9, weight_decay=0.0001)
    pre_trained_model = os.path.join(pre_snapshot_path, '{}_best_model.pth'.format(args.model))
    load_net(ema_model, pre_trained_model)
    load_net_opt(model, optimizer, pre_trained_model)
    model.train()
    ema_model.train()
    iter_num = 0
    best_performance = 0.0
    val_set = _val_set(args, device) if args.val_every > 0 else []
    while iter_num < args.max_iterations:
        for sampled_batch in batches(db_train, batch_sampler):
            volume_batch, label_batch = sampled_batch['image'], sampled_batch['label']
            img_a, img_b = volume_batch[:labeled_sub_bs], volume_batch[labeled_sub_bs:args.labeled_bs]
            uimg_a, uimg_b = volume_batch[args.labeled_bs:args.labeled_bs + unlabeled_sub_bs], volume_batch[args.labeled_bs + unlabeled_sub_bs:]
            lab_a, lab_b = label_batch[:labeled_sub_bs], label_batch[labeled_sub_bs:args.labeled_bs]
            with torch.no_grad():
                pre_a = ema_model(uimg_a)
                pre_b = ema_model(uimg_b)
                plab_a = get_ACDC_masks(pre_a, nms=1)
                plab_b = get_ACDC_masks(pre_b, nms=1)
                img_mask, loss_mask = generate_mask(img_a)
            net_input_unl = uimg_a * img_mask + img_a * (1 - img_mask)
            net_input_l = img_b * img_mask + uimg_b * (1 - img_mask)
            out_unl = model(net_input_unl)
            out_l = model(net_input_l)
            unl_dice, unl_ce = mix_loss(out_unl, plab_a, lab_a, loss_mask, u_weight=args.u_weight, unlab=True)
            l_dice, l_ce = mix_loss(out_l, lab_b, plab_b, loss_mask, u_weight=args.u_weight)
            loss_ce = unl_ce + l_ce
            loss_dice = unl_dice + l_dice
            loss = (loss_dice + loss_ce) / 2
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            iter_num += 1
            update_model_ema(model, ema_model, 0.99)
            if iter_num % args.log_every == 0:
                logging.info('iteration %d: loss: %f, mix_dice: %f, mix_ce: %f' % (iter_num, float(loss.detach()), float(loss_dice.detach()), float(loss_ce.detach())))
            if args.val_every > 0 and iter_num % args.val_every == 0:            # ACDC_BCP_train.py:402-424
                performance = _validate(model, val_set, args.num_classes)
                if performance > best_performance:
                    best_performance = performance
                    torch.save(model.state_dict(), os.path.join(snapshot_path, 'iter_{}_dice_{}.pth'.format(iter_num, round(best_performance, 4))))
                    torch.save(model.state_dict(), os.path.join(snapshot_path, '{}_best_model.pth'.format(args.model)))
                logging.info('iteration %d : mean_dice : %f' % (iter_num, performance))
            if iter_num >= args.max_iterations:
                break
    if best_performance == 0.0:
        torch.save(model.state_dict(), os.path.join(snapshot_path, '{}_best_model.pth'.format(args.model)))


def main(argv=None):
    args = parser.parse_args(argv)
    if args.deterministic:
        torch.manual_seed(args.seed)
        random.seed(args.seed)
        np.random.seed(args.seed)
    device = torch.device("cuda", torch.cuda.current_device())
    pre_snapshot_path = "./model/BCP/ACDC_{}_{}_labeled/pre_train".format(args.exp, args.labelnum)
    self_snapshot_path = "./model/BCP/ACDC_{}_{}_labeled/self_train".format(args.exp, args.labelnum)
    for snapshot_path in [pre_snapshot_path, self_snapshot_path]:
        os.makedirs(snapshot_path, exist_ok=True)
    logging.basicConfig(level=logging.INFO, format='[%(asctime)s.%(msecs)03d] %(message)s', datefmt='%H:%M:%S', stream=sys.stdout)
    logging.info(str(args))
    pre_train(args, pre_snapshot_path, device)
    self_train(args, pre_snapshot_path, self_snapshot_path, device)


if __name__ == "__main__":
    main()
