"""Counterpart of code/utils/val_2d.py (ACDC): per-volume validation of the 2-D U-Net in eval() mode.

  test_single_volume(image [1,S,X,Y], label [1,S,X,Y], model, classes, patch_size=[256,256]) -> [(dice, hd95)] * (classes-1)

Slices whose size equals patch_size (the ACDC training resolution) never leave the device: all slices of the volume go through
the net in batches (eval-mode BatchNorm is per element, so batching changes nothing), softmax + argmax is the pseudo-label kernel
(first maximum wins, as torch.argmax), per-class overlap counts are integer atomics.  Other sizes take the reference's route --
scipy.ndimage.zoom(order=0) on the host, as val_2d.py:26,35 do.  hd95 is medpy CPU code: reported as nan.
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
            if (x, y) == tuple(patch_size):
                preds = []
                for i in range(0, S, batch):
                    out = model(image[i:i + batch].unsqueeze(1))
                    out = out[0] if isinstance(out, (tuple, list)) else out
                    cl = out.permute(0, 2, 3, 1).unsqueeze(1)                   # physical [B,1,X,Y,C]
                    preds.append(ops.plabel_argmax4(cl if cl.is_contiguous() else cl.contiguous())[:, 0])
                prediction = torch.cat(preds)
            else:
                from scipy.ndimage import zoom
                img = image.cpu().numpy()
                prediction = np.zeros((S, x, y), dtype=np.uint8)
                for ind in range(S):
                    sl = zoom(img[ind], (patch_size[0] / x, patch_size[1] / y), order=0)                      # :26
                    out = model(torch.from_numpy(sl).to(device)[None, None].float())
                    out = out[0] if isinstance(out, (tuple, list)) else out
                    cl = out.permute(0, 2, 3, 1).unsqueeze(1)
                    o = ops.plabel_argmax4(cl if cl.is_contiguous() else cl.contiguous())[0, 0].cpu().numpy()
                    prediction[ind] = zoom(o, (x / patch_size[0], y / patch_size[1]), order=0)                # :35
                prediction = torch.from_numpy(prediction).to(device)
    finally:
        model.train(was_training)
    gt = label.to(torch.uint8).contiguous()
    prediction = prediction.contiguous()
    return [calculate_metric_percase(prediction, gt, i) for i in range(1, classes)]


test_single_volume.__test__ = False   # name mirrors the reference module; not a pytest test
