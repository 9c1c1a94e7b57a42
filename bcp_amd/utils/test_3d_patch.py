"""Counterpart of code/utils/test_3d_patch.py on the device: sliding-window inference of a V-Net in
eval() mode, Dice / Jaccard of the result.  The volume, the score map, the visit counts and the label map stay in HBM;
patches are views gathered into a batch, the network runs `batch` patches per call (eval-mode BatchNorm is per element,
so batching does not change a single value), the softmax + accumulation + threshold are kernels (csrc/eval.hip).

  test_single_case(model, image, stride_xy, stride_z, patch_size, num_classes=1) -> (label_map, score_map)
      (:82-141; same padding rule, same patch grid, same `score > 0.5` rule; returns device tensors)
  var_all_case_LA(model, num_classes, patch_size, stride_xy, stride_z, cases=...) -> mean Dice          (:20-38)
  calculate_metric_percase(pred, gt) -> (dice, jc, hd95, asd)   dice / jc on the device; the two surface distances need
      medpy's CPU distance transforms and are reported as nan here (out of scope).

The reference reads the LA test list from h5 files; here `cases` is an iterable of (image, label) arrays / tensors.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from ..hip_ops import Ops


def _ops_for(t):
    from . import BCP_utils as BU
    return Ops.product() if t.is_cuda else BU._cpu_ops()


def _to_dev(a, device, dtype):
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a))
    return a.to(device=device, dtype=dtype)


def sliding_window_scores(model, image, stride_xy, stride_z, patch_size, classes=(1,), batch=4, device=None):
    """the shared core of the reference's two test_single_case flavours (utils/test_3d_patch.py:82-141, pancreas/test_util.py:88-148):
    zero-pad up to the patch size, walk the patch grid, run the eval-mode net on `batch` patches per call, accumulate the softmax
    probability of every class in `classes` and the visit counts on the device.
    -> (scores: one float32 [WW,HH,DD] SUM per requested class, cnt, crop) with crop = the slices that undo the padding (or None)."""
    if device is None:
        device = next(model.parameters()).device
    image = _to_dev(image, device, torch.float32)
    w, h, d = image.shape
    pads = []
    for size, p in zip((w, h, d), patch_size):
        tot = max(p - size, 0)
        pads.append((tot // 2, tot - tot // 2))
    add_pad = any(l or r for l, r in pads)
    if add_pad:   # constant-zero padding (:104): F.pad takes the last dim first
        image = torch.nn.functional.pad(image, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    ww, hh, dd = image.shape
    sx = math.ceil((ww - patch_size[0]) / stride_xy) + 1
    sy = math.ceil((hh - patch_size[1]) / stride_xy) + 1
    sz = math.ceil((dd - patch_size[2]) / stride_z) + 1
    ops = _ops_for(image)
    scores = [torch.zeros((ww, hh, dd), dtype=torch.float32, device=device) for _ in classes]
    cnt = torch.zeros((ww, hh, dd), dtype=torch.float32, device=device)
    scratch = torch.zeros((ww, hh, dd), dtype=torch.float32, device=device) if len(classes) > 1 else None
    origins = [(min(stride_xy * x, ww - patch_size[0]), min(stride_xy * y, hh - patch_size[1]), min(stride_z * z, dd - patch_size[2]))
               for x in range(sx) for y in range(sy) for z in range(sz)]
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            for i in range(0, len(origins), batch):
                chunk = origins[i:i + batch]
                patches = torch.stack([image[xs:xs + patch_size[0], ys:ys + patch_size[1], zs:zs + patch_size[2]] for xs, ys, zs in chunk])
                out = model(patches.unsqueeze(1))
                logits = out[0] if isinstance(out, (tuple, list)) else out       # logical [B,C,X,Y,Z], physical NDHWC
                cl = logits.permute(0, 2, 3, 4, 1)
                cl = cl if cl.is_contiguous() else cl.contiguous()
                for b, org in enumerate(chunk):                                  # patches overlap: accumulate in stream order
                    for ci, c in enumerate(classes):                             # (the kernel counts a visit per call: only the first class's counts are kept)
                        ops.sw_accumulate(cl[b], scores[ci], cnt if ci == 0 else scratch, org, cls=c)
    finally:
        model.train(was_training)
    crop = tuple(slice(l, l + s) for (l, _), s in zip(pads, (w, h, d))) if add_pad else None
    return scores, cnt, crop


def test_single_case(model, image, stride_xy, stride_z, patch_size, num_classes=1, batch=4, device=None):
    """utils/test_3d_patch.py:82-141.  image: [W,H,D] numpy array or tensor.  Returns (label_map uint8 [W,H,D],
    score_map float32 [num_classes,W,H,D]) as tensors on the model's device (all channels of score_map hold the class-1
    probability, exactly as the reference's broadcast add at :131 leaves them)."""
    (score,), cnt, crop = sliding_window_scores(model, image, stride_xy, stride_z, patch_size, (1,), batch, device)
    label = _ops_for(score).sw_finish(score, cnt, 0.5)
    if crop is not None:
        label, score = label[crop].contiguous(), score[crop].contiguous()
    score_map = score.unsqueeze(0).expand(max(1, num_classes), -1, -1, -1)
    return label, score_map


def dice_jaccard(pred, gt):
    """medpy.metric.binary.dc / jc on the device: (2|A&B| / (|A|+|B|), |A&B| / |A|B|); 0.0 when the denominator is 0"""
    device = pred.device if isinstance(pred, torch.Tensor) else None
    if device is None or not isinstance(gt, torch.Tensor) or gt.device != device:
        device = pred.device if isinstance(pred, torch.Tensor) else (gt.device if isinstance(gt, torch.Tensor) else torch.device("cpu"))
    p = _to_dev(pred, device, torch.uint8).contiguous()
    g = (_to_dev(gt, device, torch.float32) != 0).to(torch.uint8).contiguous()
    c = _ops_for(p).overlap_counts(p, g, 0).tolist()     # the only host read of the validation pass
    inter, a, b = c
    dc = 2.0 * inter / (a + b) if (a + b) > 0 else 0.0
    jc = inter / (a + b - inter) if (a + b - inter) > 0 else 0.0
    return dc, jc


def calculate_metric_percase(pred, gt):
    """:180-186 -- (dice, jc, hd95, asd); the surface distances are medpy CPU code, not reproduced (nan)"""
    dc, jc = dice_jaccard(pred, gt)
    return dc, jc, float("nan"), float("nan")


def var_all_case_LA(model, num_classes, patch_size=(112, 112, 80), stride_xy=18, stride_z=4, cases=()):
    """:20-38 -- mean Dice over the cases ((image, label) pairs instead of the reference's h5 test list)"""
    total, n = 0.0, 0
    for image, label in cases:
        prediction, _ = test_single_case(model, image, stride_xy, stride_z, patch_size, num_classes=num_classes)
        if int(prediction.sum()) == 0:
            dice = 0.0
        else:
            dice = dice_jaccard(prediction, label)[0]
        total += dice
        n += 1
    return total / max(n, 1)


def test_all_case(model, cases, num_classes, patch_size=(112, 112, 80), stride_xy=18, stride_z=4, nms=0):
    """:40-80 -- per-case (dice, jc, hd95, asd), averaged; nms keeps the largest connected component (getLargestCC :11-18)"""
    tot, n = np.zeros(4), 0
    for image, label in cases:
        prediction, _ = test_single_case(model, image, stride_xy, stride_z, patch_size, num_classes=num_classes)
        if nms:
            prediction = _ops_for(prediction).cc_largest(prediction.unsqueeze(0).contiguous(), 1, 3)[0]
        m = (0.0, 0.0, 0.0, 0.0) if int(prediction.sum()) == 0 else calculate_metric_percase(prediction, label)
        tot += np.asarray(m)
        n += 1
    return tot / max(n, 1)


test_single_case.__test__ = False   # not pytest tests: names mirror the reference module
test_all_case.__test__ = False
