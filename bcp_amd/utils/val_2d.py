"""Counterpart of code/utils/val_2d.py (ACDC): per-volume validation of the 2-D U-Net in eval() mode.

  test_single_volume(image [1,S,X,Y], label [1,S,X,Y], model, classes, patch_size=[256,256]) -> [(dice, hd95)] * (classes-1)

Nothing leaves the device: slices of another size than patch_size are zoomed to it and the label maps back with the same
nearest-neighbour gather that restates scipy.ndimage.zoom(order=0) (val_2d.py:26,35; csrc/eval.hip k_acdc_augment), all slices
of the volume go through the net in batches (eval-mode BatchNorm is per element, so batching changes nothing), softmax + argmax
is the pseudo-label kernel (first maximum wins, as torch.argmax), per-class overlap counts are integer atomics.
hd95 is medpy CPU code: reported as nan.
"""
from __future__ import annotations

import numpy as np
import torch

from ..hip_ops import Ops


def _ops_for(t):
    from . import BCP_utils as BU
    return Ops.product() if t.is_cuda else BU._cpu_ops()


def calculate_metric_percase(pred_u8, gt_u8, cls):
    """:9-17 on the device: (dice, hd95) of (pred == cls, gt == cls); (0, 0) when the prediction is empty"""
    inter, a, b = _ops_for(pred_u8).overlap_counts(pred_u8, gt_u8, cls).tolist()
    if a == 0:
        return 0, 0
    dice = 2.0 * inter / (a + b) if (a + b) > 0 else 0.0
    return dice, float("nan")


def test_single_volume(image, label, model, classes, patch_size=(256, 256), batch=16):
    device = next(model.parameters()).device
    image = image.squeeze(0).to(device=device, dtype=torch.float32)
    label = label.squeeze(0).to(device)
    S, x, y = image.shape
    ops = _ops_for(image)
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            # other slice sizes: the reference zooms every slice to the training resolution and the label map back, both with
            # scipy.ndimage.zoom(order=0) (val_2d.py:26,35) -- here the same nearest-neighbour gathers on the device
            # (bcp_acdc_augment mode 0, bit-identical to scipy's order-0 zoom: tests/golden/aug_acdc.npz)
            resize = (x, y) != tuple(patch_size)
            preds = []
            for i in range(0, S, batch):
                sl = image[i:i + batch]
                if resize:
                    sl = torch.stack([ops.acdc_augment(s.contiguous(), patch_size, 0) for s in sl])
                out = model(sl.unsqueeze(1))
                out = out[0] if isinstance(out, (tuple, list)) else out
                cl = out.permute(0, 2, 3, 1).unsqueeze(1)                   # physical [B,1,X,Y,C]
                lab = ops.plabel_argmax4(cl if cl.is_contiguous() else cl.contiguous())[:, 0]
                if resize:
                    lab = torch.stack([ops.acdc_augment(o.contiguous(), (x, y), 0) for o in lab])
                preds.append(lab)
            prediction = torch.cat(preds)
    finally:
        model.train(was_training)
    gt = label.to(torch.uint8).contiguous()
    prediction = prediction.contiguous()
    return [calculate_metric_percase(prediction, gt, i) for i in range(1, classes)]


test_single_volume.__test__ = False   # name mirrors the reference module; not a pytest test
