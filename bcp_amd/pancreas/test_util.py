"""Counterpart of code/pancreas/test_util.py on the device (pancreas flavour): sliding-window inference of the
IN-V-Net and the per-case metrics that gate checkpoint selection in train_pancreas.py (:66-79, :126-141).

  test_single_case(net, image, stride_xy, stride_z, patch_size, num_classes=1, TMI=0) -> (label_map, score_map)     (:88-148)
      both softmax channels are accumulated and the label is the ARGMAX of the averaged scores (utils/test_3d_patch.py keeps one
      channel and thresholds at 0.5 instead); same zero-padding rule and patch grid.  Device tensors in, device tensors out.
  test_all_case(net, cases, num_classes, patch_size, stride_xy, stride_z, nms=0) -> (avg_metric[4], metric_list)         (:152-185)
  test_calculate_metric(net, test_dataset, num_classes=2, dim=(96, 96, 96), s_xy=18, s_z=4, nms=0) -> the same            (:188-199)
  calculate_metric_percase(pred, gt) -> (dice, jc, hd95, asd): Dice / Jaccard from integer overlap counts on the device; medpy's
      surface distances are CPU distance transforms and are reported as nan.

The reference reads h5 files named by `test_dataset.image_list`; here a dataset / `cases` is an iterable of (image, label) arrays.
"""
from __future__ import annotations

import numpy as np
import torch

from ..utils import test_3d_patch as T3


def test_single_case(net, image, stride_xy, stride_z, patch_size, num_classes=1, TMI=0, batch=4, device=None):
    (s0, s1), cnt, crop = T3.sliding_window_scores(net, image, stride_xy, stride_z, patch_size, (0, 1), batch, device)
    ops = T3._ops_for(s0)
    ops.sw_finish(s0, cnt, 0.5)            # score /= cnt in place (:142); the thresholded maps are not used
    ops.sw_finish(s1, cnt, 0.5)
    label = (s1 > s0).to(torch.uint8)      # np.argmax over two channels: the first maximum wins (:143)
    score = torch.stack([s0, s1])
    if crop is not None:
        label, score = label[crop].contiguous(), score[(slice(None),) + crop].contiguous()
    return label, score


def calculate_metric_percase(pred, gt):
    dc, jc = T3.dice_jaccard(pred, gt)
    return dc, jc, float("nan"), float("nan")


def getLargestCC(segmentation):
    """:13-17 -- skimage.measure.label default (full connectivity) + the largest component, on the device"""
    seg = segmentation.to(torch.uint8).contiguous()
    return T3._ops_for(seg).cc_largest(seg.unsqueeze(0), 1, 3)[0]


def test_all_case(net, cases, num_classes, patch_size=(112, 112, 80), stride_xy=18, stride_z=4, nms=0, TMI=0):
    total, metric_list = np.zeros(4), []
    for image, label in cases:
        prediction, _ = test_single_case(net, image, stride_xy, stride_z, patch_size, num_classes=num_classes, TMI=TMI)
        if nms:
            prediction = getLargestCC(prediction)
        if int(prediction.sum()) == 0:                                           # :168-169
            single = (0, 0, 0, 0)
        else:
            single = calculate_metric_percase(prediction, label)
        total += np.asarray(single, dtype=np.float64)
        metric_list.append(single)
    return total / max(len(metric_list), 1), metric_list


@torch.no_grad()
def test_calculate_metric(net, test_dataset, num_classes=2, dim=(96, 96, 96), s_xy=18, s_z=4, pancreas=True, DTC=False, nms=0):
    was_training = net.training
    net.eval()                                                                   # :190
    try:
        return test_all_case(net, test_dataset, num_classes=num_classes, patch_size=dim, stride_xy=s_xy, stride_z=s_z, nms=nms)
    finally:
        net.train(was_training)
