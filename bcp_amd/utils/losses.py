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
    """The reference's class (utils/losses.py:79-134), same call contract:
    `forward(inputs, target, mask=None, weight=None, softmax=False)` with `inputs` PROBABILITIES [N,C,H,W] (logits when
    softmax=True), `target` / `mask` [N,1,H,W] (any of the reference's dtypes; a `BCP_utils.BoxMask` also works as the mask),
    `weight` a per-class list.  Per-class Dice over the whole batch with squared denominators, smooth 1e-10 (masked and
    unmasked alike, utils/losses.py:94 / :105), sum(weight_i * dice_i) / n_classes.  Gradients flow to `inputs` (csrc/loss.hip: bcp_dice_prob_fwd / _bwd);
    the fused training step does not come through here (train_step.acdc_mix_loss: one pass over the logits for both terms)."""

    def __init__(self, n_classes):
        super().__init__()
        if not 2 <= int(n_classes) <= 4:
            raise ValueError("DiceLoss: n_classes in 2..4 (BCP uses 4)")
        self.n_classes = int(n_classes)

    def forward(self, inputs, target, mask=None, weight=None, softmax=False):
        if softmax:
            inputs = torch.softmax(inputs, dim=1)
        if target.dim() == inputs.dim():
            assert target.shape[1] == 1, "predict & target shape do not match"
            target = target[:, 0]
        assert inputs.shape[1] == self.n_classes and tuple(target.shape) == (inputs.shape[0],) + tuple(inputs.shape[2:]), \
            "predict & target shape do not match"
        if weight is not None:
            assert len(weight) == self.n_classes
        inputs = inputs if inputs.dtype == torch.float32 else inputs.float()
        ops = BU._ops_for(inputs)
        lab = ops.to_u8(target)
        mode, box6, m8 = 0, None, None
        if isinstance(mask, BU.BoxMask):
            mode, box6 = (3 if mask.complement else 2), mask.box6()
        elif mask is not None:
            m = mask[:, 0] if mask.dim() == inputs.dim() else mask
            m8 = ops.to_u8(m if tuple(m.shape) == tuple(target.shape) else m.expand(target.shape))
            mode = 1
        return _DiceProb.apply(inputs, lab, m8, mode, box6, None if weight is None else tuple(float(w) for w in weight))


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


def _dense(t):
    """non-overlapping and dense (any dim order): what bcp_dice_prob_* address through three strides"""
    return sum((e - 1) * st for e, st in zip(t.shape, t.stride())) + 1 == t.numel()


def _spatial_collapses(t):
    """the spatial dims of [N,C,*sp] form ONE index with stride t.stride(-1) (true for NCHW- and NHWC-contiguous)"""
    exp = t.stride(-1)
    for e, st in zip(reversed(t.shape[2:]), reversed(t.stride()[2:])):
        if e != 1 and st != exp:
            return False
        exp *= e
    return True


class _DiceProb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, lab, m8, mode, box6, weight):
        pd = probs.detach()
        if not (_dense(pd) and _spatial_collapses(pd)):
            pd = pd.contiguous()
        ops = BU._ops_for(pd)
        out, ws = ops.dice_prob_fwd(pd, lab, m8, mode, box6, weight)
        ctx.save_for_backward(pd, lab, ws)
        ctx.meta = (m8, mode, box6)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        pd, lab, ws = ctx.saved_tensors
        m8, mode, box6 = ctx.meta
        ops = BU._ops_for(pd)
        d = ops.dice_prob_bwd(pd, lab, ws, m8, mode, box6, g_dev=g.reshape(1).to(torch.float32).contiguous())
        return d, None, None, None, None, None


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
