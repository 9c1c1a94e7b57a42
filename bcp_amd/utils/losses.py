"""Drop-ins for the loss classes the train scripts use from the reference's utils/losses.py:
`mask_DiceLoss` (:8-77, LA / pancreas) and `DiceLoss` (:79-134, ACDC), on the fused HIP loss kernel
(csrc/loss.hip).  Only the dice part of the kernel's output is returned, as the reference classes do."""
import torch.nn as nn

from .. import hip_ops as H
from . import BCP_utils as BU


class mask_DiceLoss(nn.Module):
    def __init__(self, nclass, class_weights=None, smooth=1e-5):
        super().__init__()
        assert nclass == 2 and class_weights is None and smooth == 1e-5, "BCP uses mask_DiceLoss(nclass=2) with defaults"

    def forward(self, logits, target, mask=None):
        cl = BU._as_cl(logits)
        ops = BU._ops_for(cl)
        N, sp = cl.shape[0], tuple(logits.shape[2:])
        lab = BU._labels_u8(ops, target, N, sp)
        if mask is None:
            box6, m8 = (0, 0, 0, 0, 0, 0), None
        else:
            box6, m8 = BU._mask_args(mask, ops, N, sp)
        return _DiceOnly.apply(cl, lab, box6, m8, H.LOSS_LA)


class DiceLoss(nn.Module):
    """ACDC flavour; `inputs` are LOGITS here (pass softmax=... is ignored: the kernel applies the softmax)."""

    def __init__(self, n_classes):
        super().__init__()
        assert n_classes == 4

    def forward(self, logits, target, mask=None):
        cl = BU._as_cl(logits)
        ops = BU._ops_for(cl)
        N, sp = cl.shape[0], tuple(logits.shape[2:])
        lab = BU._labels_u8(ops, target.squeeze(1) if target.dim() == logits.dim() else target, N, sp)
        box6, m8 = ((0, 0, 0, 0, 0, 0), None) if mask is None else BU._mask_args(mask.squeeze(1) if hasattr(mask, "dim") and mask.dim() == logits.dim() else mask, ops, N, sp)
        return _DiceOnly.apply(cl, lab, box6, m8, H.LOSS_ACDC)


import torch  # noqa: E402


class _DiceOnly(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_cl, lab, box6, m8, flavour):
        ops = BU._ops_for(logits_cl)
        out3, ws = ops.mixloss_fwd(logits_cl, lab, lab, box6, flavour, 1.0, 0.0, mask=m8)
        ctx.save_for_backward(logits_cl, lab, ws)
        ctx.meta = (box6, m8, flavour)
        return out3[2] if flavour == H.LOSS_LA else out3[0]

    @staticmethod
    def backward(ctx, g):
        logits_cl, lab, ws = ctx.saved_tensors
        box6, m8, flavour = ctx.meta
        ops = BU._ops_for(logits_cl)
        gd = g.reshape(1).to(torch.float32)
        g_dev = torch.cat([gd, torch.zeros_like(gd)]).contiguous()
        d = ops.mixloss_bwd(logits_cl, lab, lab, box6, flavour, ws, 1.0, 1.0, mask=m8, g_dev=g_dev)
        return d, None, None, None, None


def sup_loss_parts(outputs, label):
    """(mean CE, unmasked Dice) in ONE pass -- LA_BCP_train.py:159-160 computes them with two ops"""
    cl = BU._as_cl(outputs)
    ops = BU._ops_for(cl)
    N, sp = cl.shape[0], tuple(outputs.shape[2:])
    lab = BU._labels_u8(ops, label, N, sp)
    return _CeDice.apply(cl, lab)


class _CeDice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_cl, lab):
        ops = BU._ops_for(logits_cl)
        out3, ws = ops.mixloss_fwd(logits_cl, lab, lab, (0, 0, 0, 0, 0, 0), H.LOSS_LA, 1.0, 0.0)
        ctx.save_for_backward(logits_cl, lab, ws)
        return out3[1], out3[2]

    @staticmethod
    def backward(ctx, g_ce, g_dice):
        logits_cl, lab, ws = ctx.saved_tensors
        ops = BU._ops_for(logits_cl)
        z = torch.zeros(1, dtype=torch.float32, device=logits_cl.device)
        gd = g_dice.reshape(1).float() if g_dice is not None else z
        gc = g_ce.reshape(1).float() if g_ce is not None else z
        d = ops.mixloss_bwd(logits_cl, lab, lab, (0, 0, 0, 0, 0, 0), H.LOSS_LA, ws, 1.0, 1.0, g_dev=torch.cat([gd, gc]).contiguous())
        return d, None
