"""Drop-in for the reference's utils/BCP_utils.py (the callables LA_BCP_train.py imports at :30):
`context_mask`, `mix_loss`, `sup_loss`, `update_ema_variables` -- same names, argument meaning and
results, running on the gfx950 kernels.

* `context_mask` (reference :18-28) draws the SAME three np.random.randint numbers in the same order
  but returns `BoxMask` objects instead of two dense int64 tensors: they behave like the reference's
  masks in every expression the train scripts use (`a * m + b * (1 - m)`, passing to `mix_loss`,
  `.long()`, `.sum()`), while the kernels only ever see six integers.
* `mix_loss` (:58-69) is ONE fused forward pass + ONE backward pass over the logits (csrc/loss.hip).
* `update_ema_variables` (:78-81) is one launch over the flat parameter buffer (bit-exact arithmetic).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import hip_ops as H
from ..hip_ops import Ops


class BoxMask:
    """The reference's (mask | loss_mask): ones with one zero box.  `batch` is None for the [X,Y,Z]
    image mask and N for the [N,X,Y,Z] loss mask.  `complement` marks `1 - mask`."""

    def __init__(self, box, spatial, batch=None, complement=False, device=None):
        self.box = tuple(int(v) for v in box)          # (w, h, z, pw, ph, pz) ; 2-D: (w, h, pw, ph)
        self.spatial = tuple(int(v) for v in spatial)
        self.batch = batch
        self.complement = complement
        self.device = device

    # ---- tensor-like surface used by the train scripts
    @property
    def shape(self):
        return torch.Size(((self.batch,) if self.batch is not None else ()) + self.spatial)

    def box6(self):
        if len(self.spatial) == 3:
            return self.box
        w, h, pw, ph = self.box
        return (0, w, h, 1, pw, ph)

    def long(self):
        return self

    def unsqueeze(self, _dim):
        """`mask.unsqueeze(1)` of ACDC_BCP_train.py:175-176: the kernels see the box, not a channel axis"""
        return self

    def type(self, *_a, **_k):
        return self

    def __rsub__(self, other):
        if other == 1:
            return BoxMask(self.box, self.spatial, self.batch, not self.complement, self.device)
        return NotImplemented

    def __mul__(self, t):
        return _Masked(t, self)

    __rmul__ = __mul__

    def count(self):
        vol = int(np.prod(self.spatial))
        b = int(np.prod(self.box[len(self.spatial):]))
        n = (vol - b) if not self.complement else b
        return n * (self.batch if self.batch is not None else 1)

    def sum(self):
        return torch.tensor(self.count(), dtype=torch.int64)

    def tensor(self, device=None, dtype=torch.int64):
        """materialise the dense mask (compatibility / visualisation only)"""
        m = torch.ones(self.spatial, dtype=dtype, device=device or self.device)
        if len(self.spatial) == 3:
            w, h, z, pw, ph, pz = self.box
            m[w:w + pw, h:h + ph, z:z + pz] = 0
        else:
            w, h, pw, ph = self.box
            m[w:w + pw, h:h + ph] = 0
        if self.complement:
            m = 1 - m
        if self.batch is not None:
            m = m.unsqueeze(0).repeat(self.batch, *([1] * len(self.spatial)))
        return m


class _Masked:
    """`t * mask`; adding the complementary term launches the fused copy-paste kernel"""

    def __init__(self, t, mask):
        self.t, self.mask = t, mask

    def __add__(self, other):
        if not isinstance(other, _Masked) or other.mask.box != self.mask.box or other.mask.complement == self.mask.complement:
            return self.dense() + (other.dense() if isinstance(other, _Masked) else other)
        outside, inside = (self, other) if not self.mask.complement else (other, self)
        return mix(outside.t, inside.t, self.mask)

    def dense(self):
        return self.t * self.mask.tensor(self.t.device, self.t.dtype)

    def sum(self, *a, **k):
        """`(CE(output, img_l) * mask).sum()` of ACDC_BCP_train.py:177-178 with a BoxMask: dense fallback (compatibility; the
        fused step never comes through here)"""
        return self.dense().sum(*a, **k)


def mix(a, b, mask: BoxMask, out=None):
    """a*mask + b*(1-mask) (LA_BCP_train.py:248-251): `a` outside the box, `b` inside."""
    if a.dtype != torch.float32 or a.dim() not in (4, 5) or a.shape[1] != 1:
        m = mask.tensor(a.device, a.dtype) if not mask.complement else (1 - mask).tensor(a.device, a.dtype)
        return a * m + b * (1 - m)
    ops = Ops.product() if a.is_cuda else _cpu_ops()
    N = a.shape[0]
    sp = tuple(a.shape[2:])
    cl_shape = (N,) + ((1,) + sp if len(sp) == 2 else sp) + (1,)
    o = ops.mix_box(a.contiguous().view(cl_shape), b.contiguous().view(cl_shape), mask.box6(),
                    out=None if out is None else out.view(cl_shape))
    return o.view(a.shape)


_CPU_OPS = None


def _cpu_ops():
    """CPU tensors reach the kernels only through the simulator handle that TESTS install
    (tests/conftest.py: set_test_ops); the product never has one."""
    if _CPU_OPS is None:
        raise RuntimeError("BCP HIP ops were given CPU tensors: the product path has no CPU fallback (move tensors to the GPU)")
    return _CPU_OPS


def set_test_ops(ops):
    global _CPU_OPS
    _CPU_OPS = ops


def _ops_for(t):
    return Ops.product() if t.is_cuda else _cpu_ops()


def context_mask(img, mask_ratio):
    """reference :18-28.  NB the draw bounds are hard-coded to 112/112/80 there; kept."""
    batch_size, channel, img_x, img_y, img_z = img.shape[0], img.shape[1], img.shape[2], img.shape[3], img.shape[4]
    patch_pixel_x, patch_pixel_y, patch_pixel_z = int(img_x * mask_ratio), int(img_y * mask_ratio), int(img_z * mask_ratio)
    w = np.random.randint(0, 112 - patch_pixel_x)
    h = np.random.randint(0, 112 - patch_pixel_y)
    z = np.random.randint(0, 80 - patch_pixel_z)
    box = (w, h, z, patch_pixel_x, patch_pixel_y, patch_pixel_z)
    sp = (img_x, img_y, img_z)
    return BoxMask(box, sp, None, False, img.device), BoxMask(box, sp, batch_size, False, img.device)


def _as_cl(out):
    """logical [N,C,X,Y,Z] (or [N,C,H,W]) -> physical [N,D,H,W,C] without a copy when it already is"""
    if out.dim() == 4:
        out = out.unsqueeze(2)
    cl = out.permute(0, 2, 3, 4, 1)
    return cl if cl.is_contiguous() else cl.contiguous()


def _labels_u8(ops, lab, N, sp):
    lab = ops.to_u8(lab)
    return lab.view((N,) + ((1,) + sp if len(sp) == 2 else sp))


class _MixLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_cl, img_l, patch_l, box6, mask_u8, flavour, w_img, w_patch):
        ops = _ops_for(logits_cl)
        out3, ws = ops.mixloss_fwd(logits_cl, img_l, patch_l, box6, flavour, w_img, w_patch, mask=mask_u8)
        ctx.save_for_backward(logits_cl, img_l, patch_l, ws)
        ctx.meta = (box6, mask_u8, flavour)
        if flavour == H.LOSS_LA:
            return out3[0]
        return out3[0], out3[1]

    @staticmethod
    def backward(ctx, *gs):
        logits_cl, img_l, patch_l, ws = ctx.saved_tensors
        box6, mask_u8, flavour = ctx.meta
        ops = _ops_for(logits_cl)
        if flavour == H.LOSS_LA:
            g = gs[0].reshape(1).to(torch.float32)
            g_dev = torch.cat([g, g]).contiguous()
            d = ops.mixloss_bwd(logits_cl, img_l, patch_l, box6, flavour, ws, 0.5, 0.5, mask=mask_u8, g_dev=g_dev)
        else:
            z = torch.zeros(1, dtype=torch.float32, device=logits_cl.device)
            gd = gs[0].reshape(1).to(torch.float32) if gs[0] is not None else z
            gc = gs[1].reshape(1).to(torch.float32) if gs[1] is not None else z
            d = ops.mixloss_bwd(logits_cl, img_l, patch_l, box6, flavour, ws, 1.0, 1.0, mask=mask_u8, g_dev=torch.cat([gd, gc]).contiguous())
        return d, None, None, None, None, None, None, None


class _MixLossPairFn(torch.autograd.Function):
    """Both mix_loss terms of a BCP step on the two halves of ONE logits tensor (grouped forward): the backward
    kernels write their halves of a single gradient buffer, so autograd never pads / adds slice gradients."""

    @staticmethod
    def forward(ctx, logits_cl, lab1, plab1, lab2, plab2, box6, mask_u8, flavour, w1, w2):
        ops = _ops_for(logits_cl)
        n = logits_cl.shape[0] // 2
        a, b = logits_cl[:n], logits_cl[n:]
        o1, ws1 = ops.mixloss_fwd(a, lab1, plab1, box6, flavour, w1[0], w1[1], mask=mask_u8)
        o2, ws2 = ops.mixloss_fwd(b, lab2, plab2, box6, flavour, w2[0], w2[1], mask=mask_u8)
        ctx.save_for_backward(logits_cl, lab1, plab1, lab2, plab2, ws1, ws2)
        ctx.meta = (box6, mask_u8, flavour)
        if flavour == H.LOSS_LA:
            return o1[0], o2[0]
        return o1[0], o1[1], o2[0], o2[1]

    @staticmethod
    def backward(ctx, *gs):
        logits_cl, lab1, plab1, lab2, plab2, ws1, ws2 = ctx.saved_tensors
        box6, mask_u8, flavour = ctx.meta
        ops = _ops_for(logits_cl)
        n = logits_cl.shape[0] // 2
        d = torch.empty_like(logits_cl)
        z = torch.zeros(1, dtype=torch.float32, device=logits_cl.device)

        def gv(g):
            return g.reshape(1).to(torch.float32) if g is not None else z

        if flavour == H.LOSS_LA:
            pairs = ((gv(gs[0]), gv(gs[0]), 0.5), (gv(gs[1]), gv(gs[1]), 0.5))
        else:
            pairs = ((gv(gs[0]), gv(gs[1]), 1.0), (gv(gs[2]), gv(gs[3]), 1.0))
        for half, (lab, plab, ws, (gd, gc, k)) in enumerate(((lab1, plab1, ws1, pairs[0]), (lab2, plab2, ws2, pairs[1]))):
            sl = slice(0, n) if half == 0 else slice(n, 2 * n)
            ops.mixloss_bwd(logits_cl[sl], lab, plab, box6, flavour, ws, k, k, mask=mask_u8, g_dev=torch.cat([gd, gc]).contiguous(),
                            out=d[sl])
        return (d,) + (None,) * 9


_GRAD_BUFFER = None      # (volatile_io) callable: the LOGITS tensor a loss is differentiating -> the tensor that network's backward plan reads its input
                         # from, or None.  Installed by train_step._backward for the duration of one backward call (removed in a finally).  It is
                         # a plain module global on purpose: autograd runs the backward nodes on ITS OWN thread, a thread-local installed by the
                         # caller is invisible there (the first fix of ADVICE r05 made it one and silently brought the 16 MB copy back).  What
                         # keeps two models apart is the KEY: the provider only answers for the logits tensor its own network handed out
                         # (HipNet.dout_buffer_for compares the storage pointer), not for "any tensor of that shape".


def set_grad_buffer_provider(fn):
    """the step functions hand the loss backward the student's dout_buffer_for (networks/_hipnet.py volatile_io): the logits gradient is then
    written where the backward plan reads it, no copy in between; None switches it off"""
    global _GRAD_BUFFER
    _GRAD_BUFFER = fn


def _grad_buffer(like):
    fn = _GRAD_BUFFER
    d = fn(like) if fn is not None else None
    if d is None or d.shape != like.shape or d.dtype != like.dtype or d.device != like.device:
        d = torch.empty_like(like)
    return d


class _MixLossPairTotalFn(torch.autograd.Function):
    """_MixLossPairFn whose differentiable output is the step's TOTAL loss -- LA / pancreas loss_l + loss_u, ACDC ((unl_dice + l_dice) +
    (unl_ce + l_ce)) / 2, summed by the second call's finalize in the reference's fp32 order (bcp_mixloss_fwd prev / total) -- so that no
    torch elementwise launch sits between the forward and the backward pass (round 4: the step functions' three adds, the division, their
    autograd twins and two torch.cat calls were ~10 dependent 5 us launches per step).  The individual terms come back detached, for
    logging, exactly as _MixLossPairFn computes them.  d total / d term is 1/2 for every (dice, ce) pair of both flavours."""

    @staticmethod
    def forward(ctx, logits_cl, lab1, plab1, lab2, plab2, box6, mask_u8, flavour, w1, w2):
        ops = _ops_for(logits_cl)
        n = logits_cl.shape[0] // 2
        a, b = logits_cl[:n], logits_cl[n:]
        ctx.pair = bool(ops.MIXLOSS_PAIR and logits_cl.shape[0] == 2 * n and logits_cl.is_contiguous())
        if ctx.pair:        # both calls in one launch pair (bcp_mixloss_pair_fwd): bit-identical results, three launches fewer per step
            o6, total, ws1 = ops.mixloss_pair_fwd(logits_cl, lab1, plab1, lab2, plab2, box6, flavour, w1, w2, mask=mask_u8)
            o1, o2, ws2 = o6[0], o6[1], ws1
        else:
            total = torch.empty(1, dtype=torch.float32, device=logits_cl.device)
            o1, ws1 = ops.mixloss_fwd(a, lab1, plab1, box6, flavour, w1[0], w1[1], mask=mask_u8)
            o2, ws2 = ops.mixloss_fwd(b, lab2, plab2, box6, flavour, w2[0], w2[1], mask=mask_u8, prev=o1, total=total)
        ctx.save_for_backward(logits_cl, lab1, plab1, lab2, plab2, ws1, ws2)
        ctx.meta = (box6, mask_u8, flavour)
        terms = (o1[0], o2[0]) if flavour == H.LOSS_LA else (o1[0], o1[1], o2[0], o2[1])
        ctx.mark_non_differentiable(*terms)
        ctx.set_materialize_grads(False)       # (the terms' gradients are never used: materialised as zeros they were one 1-workgroup fill launch
                                               #  each between the forward and the backward pass -- 13 us of the step's critical path at the LA size)
        return (total.reshape(()),) + terms

    @staticmethod
    def backward(ctx, g, *unused):
        logits_cl, lab1, plab1, lab2, plab2, ws1, ws2 = ctx.saved_tensors
        box6, mask_u8, flavour = ctx.meta
        ops = _ops_for(logits_cl)
        n = logits_cl.shape[0] // 2
        if g is None:                          # (set_materialize_grads(False): the total was not part of what is being differentiated)
            return (None,) * 10
        d = _grad_buffer(logits_cl)
        g1 = g.reshape(1)
        if g1.dtype != torch.float32 or not g1.is_contiguous():
            g1 = g1.to(torch.float32).contiguous()
        if ctx.pair:
            ops.mixloss_pair_bwd(logits_cl, lab1, plab1, lab2, plab2, box6, flavour, ws1, 0.5, 0.5, mask=mask_u8, g_dev=g1, out=d)
            return (d,) + (None,) * 9
        for half, (lab, plab, ws) in enumerate(((lab1, plab1, ws1), (lab2, plab2, ws2))):
            sl = slice(0, n) if half == 0 else slice(n, 2 * n)
            ops.mixloss_bwd(logits_cl[sl], lab, plab, box6, flavour, ws, 0.5, 0.5, mask=mask_u8, g_dev=g1, out=d[sl])
        return (d,) + (None,) * 9


def mix_loss_pair(out, first, second, mask, flavour=H.LOSS_LA, total=False):
    """`out` = logits of a grouped forward over [batch1; batch2]; `first` / `second` = (img_l, patch_l, w_img, w_patch)
    for each half.  LA: returns (loss_1, loss_2); ACDC: (dice_1, ce_1, dice_2, ce_2).  total=True: (total, *terms) with `total` the
    step's loss as the reference sums it and the only differentiable output (_MixLossPairTotalFn)."""
    cl = _as_cl(out)
    ops = _ops_for(cl)
    N, sp = cl.shape[0] // 2, tuple(out.shape[2:])
    box6, m8 = _mask_args(mask, ops, N, sp)
    l1, p1, w1a, w1b = first
    l2, p2, w2a, w2b = second
    fn = _MixLossPairTotalFn if total else _MixLossPairFn
    return fn.apply(cl, _labels_u8(ops, l1, N, sp), _labels_u8(ops, p1, N, sp), _labels_u8(ops, l2, N, sp),
                    _labels_u8(ops, p2, N, sp), box6, m8, flavour, (float(w1a), float(w1b)), (float(w2a), float(w2b)))


_ONES = {}


def unit_gradient(like):
    """a cached 0-dim float32 one on `like`'s device: `loss.backward(gradient=unit_gradient(loss))` spares the fill launch autograd would
    make for the implicit gradient of a scalar loss"""
    k = (like.device.type, like.device.index)
    o = _ONES.get(k)
    if o is None:
        o = _ONES[k] = torch.ones((), dtype=torch.float32, device=like.device)
    return o


def _mask_args(mask, ops, N, sp):
    """-> (box6, dense uint8 mask or None)"""
    if isinstance(mask, BoxMask):
        if mask.complement:
            raise ValueError("pass the loss mask itself, not its complement")
        return mask.box6(), None
    m = ops.to_u8(mask if mask.dim() == len(sp) + 1 else mask.expand((N,) + sp))
    return (0, 0, 0, 0, 0, 0), m.view((N,) + ((1,) + sp if len(sp) == 2 else sp))


def mix_loss(net3_output, img_l, patch_l, mask, l_weight=1.0, u_weight=0.5, unlab=False):
    """reference :58-69 (LA) / pancreas/losses.py:129-141"""
    image_weight, patch_weight = l_weight, u_weight
    if unlab:
        image_weight, patch_weight = u_weight, l_weight
    cl = _as_cl(net3_output)
    ops = _ops_for(cl)
    N, sp = cl.shape[0], tuple(net3_output.shape[2:])
    box6, m8 = _mask_args(mask, ops, N, sp)
    return _MixLossFn.apply(cl, _labels_u8(ops, img_l, N, sp), _labels_u8(ops, patch_l, N, sp), box6, m8, H.LOSS_LA,
                            float(image_weight), float(patch_weight))


def sup_loss(output, label):
    """reference :71-76: (unmasked dice + mean CE) / 2 == mix_loss with an empty box and weight 1"""
    cl = _as_cl(output)
    ops = _ops_for(cl)
    N, sp = cl.shape[0], tuple(output.shape[2:])
    lab = _labels_u8(ops, label, N, sp)
    return _MixLossFn.apply(cl, lab, lab, (0, 0, 0, 0, 0, 0), None, H.LOSS_LA, 1.0, 0.0)


@torch.no_grad()
def update_ema_variables(model, ema_model, alpha):
    """reference :78-81: ema = alpha*ema + (1-alpha)*param over .parameters() -- one launch when both
    models keep their parameters in a flat buffer (networks/_hipnet.py), else one launch per tensor."""
    from ..networks._hipnet import HipNet
    if isinstance(model, HipNet) and isinstance(ema_model, HipNet):
        src, dst = model.flat_params(), ema_model.flat_params()
        if src.numel() == dst.numel():
            _ops_for(dst).ema(dst, src, alpha)
            ema_model.bump()
            return
    for ema_param, param in zip(ema_model.parameters(), model.parameters()):
        _ops_for(param).ema(ema_param.data.view(-1), param.data.contiguous().view(-1), alpha)
