"""The BCP self-training step as the reference's loops run it (LA_BCP_train.py:233-276,
pancreas/train_pancreas.py:144-171, ACDC_BCP_train.py:353-390), expressed over the drop-in API.
bench.py, the train scripts and the parity tests all call these functions, so the thing that is timed
is the thing that is tested.
"""
from __future__ import annotations

import numpy as np
import torch

from . import hip_ops as H
from .hip_ops import Ops
from .utils import BCP_utils as BU


# grouped steps: the second mix_loss launch leaves the step's total loss on the device (BU.mix_loss_pair(total=True)) and the backward pass
# starts from a cached unit gradient -- no torch elementwise launches between forward and backward.  False: the reference's adds / division
# as torch ops (measurement switch, bench.py --opt step_total=0)
STEP_TOTAL = True


def _ops_for(t):
    return BU._ops_for(t)


_SIDE = {}


TEACHER_STREAM_PRIORITY = 0      # measurement switch (bench.py --opt teacher_prio=-1): HIP priority of the teacher's side stream


def _backward(model, loss):
    """loss.backward() with the student's backward-plan input tensor offered to the loss backward (volatile_io, networks/_hipnet.py)"""
    vol = getattr(model, "volatile_io", False)
    if vol:
        BU.set_grad_buffer_provider(model.dout_buffer_for)
    try:
        loss.backward(gradient=BU.unit_gradient(loss))
    finally:
        if vol:
            BU.set_grad_buffer_provider(None)


def _side_stream(t):
    s = _SIDE.get(t.device)
    if s is None:
        s = _SIDE[t.device] = torch.cuda.Stream(device=t.device, priority=TEACHER_STREAM_PRIORITY)
    return s


# ------------------------------------------------------------------------------------------ pseudo labels
PLABEL_CC_FUSED = True    # (round 6) pseudo-label + largest-CC as one chain (bcp_plabel_cc_largest); False: bcp_plabel_* then bcp_cc_largest (rounds 1-5; bench.py --opt plabel_cc_fused=0)
def get_cut_mask(out, thres=0.5, nms=0, connect_mode=None):
    """LA_BCP_train.py:57-63 / pancreas_utils.py:275-281: softmax -> (p>=thres) -> channel 1 [-> largest CC].
    Returns uint8 [N,X,Y,Z] on the device (the reference returns int64 / float32 of the same values);
    connect_mode None = full connectivity (skimage default, 26), 2 = 18, 1 = 6."""
    cl = BU._as_cl(out)
    ops = _ops_for(cl)
    conn = {None: 3, 3: 3, 2: 2, 1: 1}[connect_mode]
    if nms and PLABEL_CC_FUSED:      # (round 6) one chain: the first largest-CC kernel labels from the logits itself (bcp_plabel_cc_largest)
        return ops.plabel_cc_largest(cl, thres, conn)
    seg = ops.plabel_bin(cl, thres)
    return ops.cc_largest(seg, 1, conn) if nms else seg


def get_ACDC_masks(output, nms=0):
    """ACDC_BCP_train.py:112-117: softmax -> argmax [-> per-class largest 8-connected component] -> uint8 [N,H,W]"""
    cl = BU._as_cl(output)
    ops = _ops_for(cl)
    if nms and PLABEL_CC_FUSED:
        seg = ops.plabel_cc_largest(cl, 0.5, 2)          # [N,1,H,W]
    else:
        seg = ops.plabel_argmax4(cl)
        if nms:
            seg = ops.cc_largest(seg, 3, 2)
    return seg.view(seg.shape[0], seg.shape[2], seg.shape[3])


# ------------------------------------------------------------------------------------------ optimisers
def _opt_param_slices(model):
    """[(index in model.parameters(), flat offset, parameter)] of the parameters the optimiser steps (the flat trainable
    prefix).  Indices follow `model.parameters()` -- the order torch.optim numbers them in a state_dict -- so an 'opt' entry
    written by the reference's save_net_opt (LA_BCP_train.py:79-84, ACDC_BCP_train.py:60-64, pancreas_utils.py:160-168)
    maps onto the flat buffers and back."""
    model._ensure_flat()
    out = []
    for i, q in enumerate(model._ordered_params()):
        if id(q) in model._opt_param_ids:
            out.append((i, model._offs[id(q)], q))
    return out


def _n_params(model):
    return len(model._ordered_params())


class FlatSGD:
    """torch.optim.SGD(momentum, weight_decay) semantics (LA_BCP_train.py:218) as ONE launch over the flat
    trainable buffer of a HipNet; parameters whose grad is None in the reference (the unused heads) are
    outside that buffer, exactly as torch skips them."""

    def __init__(self, model, lr=0.01, momentum=0.9, weight_decay=1e-4):
        self.model = model
        self.param_groups = [{"lr": lr, "momentum": momentum, "weight_decay": weight_decay}]
        self.buf = None
        self.steps = 0
        self.grad_scale = 1.0

    def zero_grad(self, set_to_none=True):
        # the reference drops / zeroes 150 gradient tensors here; ours all live in one flat buffer that the next backward
        # clears with a single memset (HipNet.begin_backward) -- nothing to do per parameter
        self.model.mark_grads_stale()

    def step(self):
        g = self.param_groups[0]
        p, gr = self.model.flat_trainable()
        if self.buf is None:
            self.buf = torch.zeros_like(p)
        _ops_for(p).sgd(p, gr, self.buf, g["lr"], g["momentum"], g["weight_decay"], first_step=(self.steps == 0), grad_scale=self.grad_scale)
        self.steps += 1
        self.model.bump()

    def state_dict(self):
        """torch.optim.SGD's layout: {'state': {i: {'momentum_buffer': tensor}}, 'param_groups': [{..., 'params': [0..n-1]}]}
        with per-parameter VIEWS of the flat momentum buffer (no state before the first step, as in torch)."""
        g = self.param_groups[0]
        state = {}
        if self.buf is not None and self.steps > 0:
            for i, off, q in _opt_param_slices(self.model):
                state[i] = {"momentum_buffer": self.buf[off:off + q.numel()].view(q.shape)}
        group = {"lr": g["lr"], "momentum": g["momentum"], "dampening": 0, "weight_decay": g["weight_decay"], "nesterov": False,
                 "maximize": False, "foreach": None, "differentiable": False, "fused": None, "params": list(range(_n_params(self.model)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """accepts torch.optim.SGD's state_dict (the reference's checkpoints) -- and, for files written by round 1 of this
        package, the old private {'buf','steps','param_groups'} layout"""
        if "state" not in sd:
            self.buf, self.steps = sd["buf"], sd["steps"]
            self.param_groups = [dict(sd["param_groups"][0])]
            return
        g = sd["param_groups"][0]
        self.param_groups = [{"lr": g["lr"], "momentum": g["momentum"], "weight_decay": g["weight_decay"]}]
        p, _ = self.model.flat_trainable()
        self.buf = torch.zeros_like(p)
        self.steps = 0
        st = sd["state"]
        for i, off, q in _opt_param_slices(self.model):
            e = st.get(i, st.get(str(i)))
            if e is not None and e.get("momentum_buffer") is not None:
                self.buf[off:off + q.numel()].copy_(e["momentum_buffer"].reshape(-1).to(self.buf.device, torch.float32))
                self.steps = max(self.steps, 1)      # momentum is live: the next step is not a "first step"


class FlatAdam:
    """torch.optim.Adam(lr) defaults (pancreas/dataloaders.py:182) over the flat trainable buffer"""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps}]
        self.m = self.v = None
        self.steps = 0
        self.grad_scale = 1.0

    def zero_grad(self, set_to_none=True):
        # the reference drops / zeroes 150 gradient tensors here; ours all live in one flat buffer that the next backward
        # clears with a single memset (HipNet.begin_backward) -- nothing to do per parameter
        self.model.mark_grads_stale()

    def step(self):
        g = self.param_groups[0]
        p, gr = self.model.flat_trainable()
        if self.m is None:
            self.m, self.v = torch.zeros_like(p), torch.zeros_like(p)
        self.steps += 1
        _ops_for(p).adam(p, gr, self.m, self.v, g["lr"], self.steps, g["betas"][0], g["betas"][1], g["eps"], grad_scale=self.grad_scale)
        self.model.bump()

    def state_dict(self):
        """torch.optim.Adam's layout: state[i] = {'step', 'exp_avg', 'exp_avg_sq'} (views of the flat moment buffers)"""
        g = self.param_groups[0]
        state = {}
        if self.m is not None and self.steps > 0:
            for i, off, q in _opt_param_slices(self.model):
                state[i] = {"step": torch.tensor(float(self.steps)), "exp_avg": self.m[off:off + q.numel()].view(q.shape),
                            "exp_avg_sq": self.v[off:off + q.numel()].view(q.shape)}
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(_n_params(self.model)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """torch.optim.Adam's state_dict (copied: the reference loads ONE checkpoint's optimiser state into two optimisers,
        train_pancreas.py:116-117), or round 1's private {'m','v','steps','param_groups'} layout"""
        if "state" not in sd:
            self.m = None if sd["m"] is None else sd["m"].clone()
            self.v = None if sd["v"] is None else sd["v"].clone()
            self.steps, self.param_groups = sd["steps"], [dict(sd["param_groups"][0])]
            return
        g = sd["param_groups"][0]
        self.param_groups = [{"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"]}]
        p, _ = self.model.flat_trainable()
        self.m, self.v = torch.zeros_like(p), torch.zeros_like(p)
        self.steps = 0
        st = sd["state"]
        for i, off, q in _opt_param_slices(self.model):
            e = st.get(i, st.get(str(i)))
            if e is not None:
                self.m[off:off + q.numel()].copy_(e["exp_avg"].reshape(-1).to(self.m.device, torch.float32))
                self.v[off:off + q.numel()].copy_(e["exp_avg_sq"].reshape(-1).to(self.v.device, torch.float32))
                self.steps = max(self.steps, int(float(e["step"])))


# ------------------------------------------------------------------------------------------ LA / pancreas step
class _NoVolatileIO:
    """the unfused loops (grouped=False: the reference's four separate network calls) hold the first call's outputs across the second call
    of the same network, which volatile_io (networks/_hipnet.py) does not allow: those steps run with defensive copies"""

    def __init__(self, *nets):
        self.nets = [n for n in nets if getattr(n, "volatile_io", False)]

    def __enter__(self):
        for n in self.nets:
            n.volatile_io = False

    def __exit__(self, *a):
        for n in self.nets:
            n.volatile_io = True


def la_self_train_step(model, ema_model, optimizer, volume_batch, label_batch, labeled_bs, box=None, drops=None,
                       u_weight=0.5, mask_ratio=2 / 3, alpha=0.99, variant="la", connect_mode=None, dp=None, grouped=True,
                       overlap=True, plabs=None):
    if not grouped and (getattr(model, "volatile_io", False) or getattr(ema_model, "volatile_io", False)):
        with _NoVolatileIO(model, ema_model):
            return la_self_train_step(model, ema_model, optimizer, volume_batch, label_batch, labeled_bs, box, drops, u_weight, mask_ratio, alpha,
                                      variant, connect_mode, dp, grouped, overlap, plabs)
    return _la_self_train_step(model, ema_model, optimizer, volume_batch, label_batch, labeled_bs, box, drops, u_weight, mask_ratio, alpha, variant,
                               connect_mode, dp, grouped, overlap, plabs)


def _la_self_train_step(model, ema_model, optimizer, volume_batch, label_batch, labeled_bs, box=None, drops=None,
                        u_weight=0.5, mask_ratio=2 / 3, alpha=0.99, variant="la", connect_mode=None, dp=None, grouped=True,
                        overlap=True, plabs=None):
    """One self-training iteration, LA_BCP_train.py:235-270 (variant 'pancreas': train_pancreas.py:145-171).

    volume_batch [B,1,X,Y,Z] float32 laid out lab_a|lab_b|unlab_a|unlab_b, label_batch [B,X,Y,Z].
    box: explicit (w,h,z,pw,ph,pz) for parity runs, else drawn by context_mask from np.random as the
    reference does.  drops: optional injected Dropout3d keep-masks {'t_a','t_b','s_l','s_u'}.
    plabs: parity hook like `box` / `drops` -- (plab_a, plab_b) uint8 pseudo-labels the student is trained on INSTEAD of the
    teacher's (which are still computed and returned as 'plab_a' / 'plab_b'): takes the discrete pseudo-label bifurcations out
    of a multi-step comparison (tests/net_checks.py:check_la_traj5).
    dp: optional bcp_amd.dp.DataParallel (gradient all-reduce before the optimiser step).
    grouped: launch the two teacher batches (and the two student batches) as ONE grouped forward each -- separately
    normalised exactly like the reference's two calls, but half the launches (False: two calls, as the scripts read).
    Returns device scalars; nothing here synchronises with the host."""
    sub_bs = int(labeled_bs / 2)
    img_a, img_b = volume_batch[:sub_bs], volume_batch[sub_bs:labeled_bs]
    lab_a, lab_b = label_batch[:sub_bs], label_batch[sub_bs:labeled_bs]
    unimg_a, unimg_b = volume_batch[labeled_bs:labeled_bs + sub_bs], volume_batch[labeled_bs + sub_bs:]
    drops = drops or {}

    def cat_drops(k1, k2):
        d1, d2 = drops.get(k1), drops.get(k2)
        if d1 is None and d2 is None:
            return None
        return {k: torch.cat([d1[k], d2[k]]) for k in d1}

    # The student's INPUTS do not depend on the pseudo-labels (only its loss does), so the teacher forward + pseudo-label
    # + largest-CC chain runs on a side stream underneath the copy-paste mix and the student forward.
    side = _side_stream(volume_batch) if (grouped and overlap and volume_batch.is_cuda) else None
    with torch.no_grad():
        if grouped:
            # the two teacher batches are adjacent in volume_batch: ONE grouped forward, separately normalised
            ema_model.drop_masks = cat_drops("t_a", "t_b")
            if side is not None:
                main = torch.cuda.current_stream(volume_batch.device)
                side.wait_stream(main)               # last step's EMA (and this step's inputs) are ordered before the teacher
                with torch.cuda.stream(side):
                    unout = ema_model(volume_batch[labeled_bs:], groups=2, features=False)[0]
                    plab = get_cut_mask(unout, nms=1, connect_mode=connect_mode)
                plab.record_stream(main)
            else:
                unout = ema_model(volume_batch[labeled_bs:], groups=2, features=False)[0]
                plab = get_cut_mask(unout, nms=1, connect_mode=connect_mode)
            plab_a, plab_b = plab[:sub_bs], plab[sub_bs:]
        else:
            ema_model.drop_masks = drops.get("t_a")
            unoutput_a = ema_model(unimg_a, features=False)[0]
            ema_model.drop_masks = drops.get("t_b")
            unoutput_b = ema_model(unimg_b, features=False)[0]
            plab_a = get_cut_mask(unoutput_a, nms=1, connect_mode=connect_mode)
            plab_b = get_cut_mask(unoutput_b, nms=1, connect_mode=connect_mode)
        if box is None:
            if variant == "la":
                img_mask, loss_mask = BU.context_mask(img_a, mask_ratio)
            else:
                from .pancreas.pancreas_utils import generate_mask
                img_mask, loss_mask = generate_mask(img_a, 64)
        else:
            sp = tuple(volume_batch.shape[2:])
            img_mask = BU.BoxMask(box, sp, None, False, volume_batch.device)
            loss_mask = BU.BoxMask(box, sp, sub_bs, False, volume_batch.device)
    own_plabs = (plab_a, plab_b)
    if plabs is not None:
        plab_a, plab_b = plabs[0].to(volume_batch.device), plabs[1].to(volume_batch.device)
    if grouped and label_batch.is_cuda and label_batch.dtype != torch.uint8:
        # the loss reads uint8 label maps; the labeled half is converted HERE, in front of the student's forward (one launch under the
        # teacher's pass), not by the loss between the forward and the backward pass (two launches on the step's critical path)
        lab8 = _ops_for(volume_batch).to_u8(label_batch[:labeled_bs])
        lab_a, lab_b = lab8[:sub_bs], lab8[sub_bs:labeled_bs]
    # direction tables: LA_BCP_train.py:248-251 / train_pancreas.py:155-156
    pairs = ((img_a, unimg_a), (unimg_b, img_b)) if variant == "la" else ((unimg_a, img_b), (img_a, unimg_b))
    if variant == "la":
        terms = ((lab_a, plab_a, 1.0, u_weight), (plab_b, lab_b, u_weight, 1.0))       # mix_loss(.., u_weight) / (.., unlab=True)
    else:
        terms = ((plab_a, lab_b, u_weight, 1.0), (lab_a, plab_b, 1.0, u_weight))       # train_pancreas.py:160,164 (the reference passes no u_weight: mix_loss's default 0.5 = ours)
    early_zero = optimizer is not None and grouped and isinstance(optimizer, (FlatSGD, FlatAdam)) and optimizer.model is model
    if early_zero:
        # optimizer.zero_grad() of the reference's loop body, moved in front of the forward pass (nothing reads a gradient in between): its
        # one memset then runs under the teacher's pass instead of between the loss and the backward pass
        optimizer.zero_grad()
        model.clear_grads_now()
    if grouped:
        mshape = (2 * sub_bs,) + tuple(volume_batch.shape[1:])
        mixed = model.input_buffer(mshape) if getattr(model, "volatile_io", False) else None      # the forward plan's own input tensor: no copy
        if mixed is None or mixed.dtype != volume_batch.dtype:
            mixed = torch.empty(mshape, dtype=volume_batch.dtype, device=volume_batch.device)
        BU.mix(pairs[0][0], pairs[0][1], img_mask, out=mixed[:sub_bs])
        BU.mix(pairs[1][0], pairs[1][1], img_mask, out=mixed[sub_bs:])
        model.drop_masks = cat_drops("s_l", "s_u")
        outputs = model(mixed, groups=2, features=False)[0]
        if side is not None:
            torch.cuda.current_stream(volume_batch.device).wait_stream(side)   # pseudo-labels are needed from here on
        if STEP_TOTAL:
            loss, loss_l, loss_u = BU.mix_loss_pair(outputs, terms[0], terms[1], loss_mask, total=True)      # loss = loss_l + loss_u, summed on the device
        else:
            loss_l, loss_u = BU.mix_loss_pair(outputs, terms[0], terms[1], loss_mask)
            loss = loss_l + loss_u
        outputs_l, outputs_u = outputs[:sub_bs], outputs[sub_bs:]
    else:
        mixl_img = pairs[0][0] * img_mask + pairs[0][1] * (1 - img_mask)
        mixu_img = pairs[1][0] * img_mask + pairs[1][1] * (1 - img_mask)
        model.drop_masks = drops.get("s_l")
        outputs_l = model(mixl_img, features=False)[0]
        loss_l = BU.mix_loss(outputs_l, terms[0][0], terms[0][1], loss_mask, l_weight=terms[0][2], u_weight=terms[0][3])
        model.drop_masks = drops.get("s_u")
        outputs_u = model(mixu_img, features=False)[0]
        loss_u = BU.mix_loss(outputs_u, terms[1][0], terms[1][1], loss_mask, l_weight=terms[1][2], u_weight=terms[1][3])
        loss = loss_l + loss_u
    if optimizer is None:      # gradient-only mode (DP equivalence tests): caller owns zero_grad / step / EMA
        _backward(model, loss)
    else:
        if not early_zero:
            optimizer.zero_grad()
        if dp is not None and grouped:
            dp.arm(model)              # ONE backward in this step: gradient buckets go out underneath it
        _backward(model, loss)
        if dp is not None:
            dp.allreduce_grads(model, optimizer)
        optimizer.step()
        BU.update_ema_variables(model, ema_model, alpha)
    model.drop_masks = None
    ema_model.drop_masks = None
    # ALIASING CONTRACT (volatile_io, networks/_hipnet.py): with model.volatile_io set, outputs_l / outputs_u are views of the forward plan's
    # own logits tensor -- valid until this network's NEXT pass, which overwrites them in place; a caller that keeps them across steps clones
    # them.  With volatile_io off (the default) they are private copies.
    return dict(loss=loss.detach(), loss_l=loss_l.detach(), loss_u=loss_u.detach(), plab_a=own_plabs[0], plab_b=own_plabs[1],
                outputs_l=outputs_l.detach(), outputs_u=outputs_u.detach())


def la_pre_train_step(model, optimizer, volume_batch, label_batch, mask_ratio=2 / 3, box=None, variant="la"):
    """One pre-training iteration, LA_BCP_train.py:150-167 (variant 'pancreas': train_pancreas.py:83-97, a 64^3 box): the two
    halves of the LABELED batch are copy-pasted into each other (images and labels alike), the loss is the supervised
    (CE + Dice) / 2.  Returns device scalars."""
    from .utils.losses import sup_loss_parts
    sub_bs = volume_batch.shape[0] // 2
    img_a, img_b = volume_batch[:sub_bs], volume_batch[sub_bs:2 * sub_bs]
    lab_a, lab_b = label_batch[:sub_bs], label_batch[sub_bs:2 * sub_bs]
    with torch.no_grad():
        if box is None and variant == "pancreas":
            from .pancreas.pancreas_utils import generate_mask as pancreas_mask
            img_mask, _ = pancreas_mask(img_a, 64)
        elif box is None:
            img_mask, _ = BU.context_mask(img_a, mask_ratio)
        else:
            img_mask = BU.BoxMask(box, tuple(volume_batch.shape[2:]), None, False, volume_batch.device)
    mixed_img = img_a * img_mask + img_b * (1 - img_mask)
    mixed_lab = lab_a * img_mask + lab_b * (1 - img_mask)
    outputs = model(mixed_img, features=False)[0]
    loss_ce, loss_dice = sup_loss_parts(outputs, mixed_lab)
    loss = (loss_ce + loss_dice) / 2
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return dict(loss=loss.detach(), loss_ce=loss_ce.detach(), loss_dice=loss_dice.detach())


# ------------------------------------------------------------------------------------------ ACDC
def generate_mask(img):
    """ACDC_BCP_train.py:131-140: 2/3 x 2/3 zero box, two np.random.randint draws (w then h)."""
    batch_size, channel, img_x, img_y = img.shape[0], img.shape[1], img.shape[2], img.shape[3]
    patch_x, patch_y = int(img_x * 2 / 3), int(img_y * 2 / 3)
    w = np.random.randint(0, img_x - patch_x)
    h = np.random.randint(0, img_y - patch_y)
    box = (w, h, patch_x, patch_y)
    return BU.BoxMask(box, (img_x, img_y), None, False, img.device), BU.BoxMask(box, (img_x, img_y), batch_size, False, img.device)


def acdc_mix_loss(output, img_l, patch_l, mask, l_weight=1.0, u_weight=0.5, unlab=False):
    """ACDC_BCP_train.py:167-179 -> (loss_dice, loss_ce)"""
    image_weight, patch_weight = l_weight, u_weight
    if unlab:
        image_weight, patch_weight = u_weight, l_weight
    cl = BU._as_cl(output)
    ops = _ops_for(cl)
    N, sp = cl.shape[0], tuple(output.shape[2:])
    box6, m8 = BU._mask_args(mask, ops, N, sp)
    return BU._MixLossFn.apply(cl, BU._labels_u8(ops, img_l, N, sp), BU._labels_u8(ops, patch_l, N, sp), box6, m8, H.LOSS_ACDC,
                               float(image_weight), float(patch_weight))


@torch.no_grad()
def update_model_ema(model, ema_model, alpha):
    """ACDC_BCP_train.py:123-129: EMA over the whole state_dict (parameters AND BN buffers).  One launch over
    the flat state; the int64 num_batches_tracked goes through float32 and is truncated, as load_state_dict does."""
    src, dst = model.flat_state(), ema_model.flat_state()
    _ops_for(dst).ema(dst, src, alpha)
    ema_model.bump()
    a, b = float(getattr(ema_model, "_nbt", 0)), float(getattr(model, "_nbt", 0))
    ema_model._nbt = int(np.float32(np.float32(alpha) * np.float32(a)) + np.float32(np.float32(1 - alpha) * np.float32(b)))
    ema_model._nbt_dirty = True


def acdc_self_train_step(model, ema_model, optimizer, volume_batch, label_batch, labeled_bs, box=None, drops=None,
                         u_weight=0.5, alpha=0.99, dp=None, grouped=True, overlap=True, plabs=None):
    if not grouped and (getattr(model, "volatile_io", False) or getattr(ema_model, "volatile_io", False)):
        with _NoVolatileIO(model, ema_model):
            return acdc_self_train_step(model, ema_model, optimizer, volume_batch, label_batch, labeled_bs, box, drops, u_weight, alpha, dp, grouped,
                                        overlap, plabs)
    return _acdc_self_train_step(model, ema_model, optimizer, volume_batch, label_batch, labeled_bs, box, drops, u_weight, alpha, dp, grouped, overlap,
                                 plabs)


def _acdc_self_train_step(model, ema_model, optimizer, volume_batch, label_batch, labeled_bs, box=None, drops=None,
                          u_weight=0.5, alpha=0.99, dp=None, grouped=True, overlap=True, plabs=None):
    """One ACDC self-training iteration, ACDC_BCP_train.py:355-390 (grouped: see la_self_train_step; needs
    labeled_bs == batch - labeled_bs so that both halves have equal size).  plabs / optimizer=None: the parity hooks of
    la_self_train_step (forced pseudo-labels; gradient-only mode)."""
    bs = volume_batch.shape[0]
    lsub, usub = int(labeled_bs / 2), int((bs - labeled_bs) / 2)
    grouped = grouped and lsub == usub
    img_a, img_b = volume_batch[:lsub], volume_batch[lsub:labeled_bs]
    uimg_a, uimg_b = volume_batch[labeled_bs:labeled_bs + usub], volume_batch[labeled_bs + usub:]
    lab_a, lab_b = label_batch[:lsub], label_batch[lsub:labeled_bs]
    drops = drops or {}

    def cat_drops(k1, k2):
        d1, d2 = drops.get(k1), drops.get(k2)
        if d1 is None and d2 is None:
            return None
        return {k: torch.cat([d1[k], d2[k]]) for k in d1}

    side = _side_stream(volume_batch) if (grouped and overlap and volume_batch.is_cuda) else None
    with torch.no_grad():
        if grouped:
            ema_model.drop_masks = cat_drops("t_a", "t_b")
            if side is not None:     # teacher forward -> pseudo-label -> per-class CC underneath the student forward (see la_self_train_step)
                main = torch.cuda.current_stream(volume_batch.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    pre = ema_model(volume_batch[labeled_bs:], groups=2)
                    plab = get_ACDC_masks(pre, nms=1)
                plab.record_stream(main)
            else:
                pre = ema_model(volume_batch[labeled_bs:], groups=2)
                plab = get_ACDC_masks(pre, nms=1)
            plab_a, plab_b = plab[:usub], plab[usub:]
        else:
            ema_model.drop_masks = drops.get("t_a")
            pre_a = ema_model(uimg_a)
            ema_model.drop_masks = drops.get("t_b")
            pre_b = ema_model(uimg_b)
            plab_a = get_ACDC_masks(pre_a, nms=1)
            plab_b = get_ACDC_masks(pre_b, nms=1)
        if box is None:
            img_mask, loss_mask = generate_mask(img_a)
        else:
            sp = tuple(volume_batch.shape[2:])
            img_mask, loss_mask = BU.BoxMask(box, sp, None, False, volume_batch.device), BU.BoxMask(box, sp, lsub, False, volume_batch.device)
    own_plabs = (plab_a, plab_b)
    if plabs is not None:
        plab_a, plab_b = plabs[0].to(volume_batch.device), plabs[1].to(volume_batch.device)
    if grouped and label_batch.is_cuda and label_batch.dtype != torch.uint8:
        lab8 = _ops_for(volume_batch).to_u8(label_batch[:labeled_bs])      # in front of the student's forward, not between forward and backward (see la_self_train_step)
        lab_a, lab_b = lab8[:lsub], lab8[lsub:labeled_bs]
    early_zero = optimizer is not None and grouped and isinstance(optimizer, (FlatSGD, FlatAdam)) and optimizer.model is model
    if early_zero:
        optimizer.zero_grad()          # (as la_self_train_step: the memset in front of the forward pass)
        model.clear_grads_now()
    if grouped:
        mshape = (2 * lsub,) + tuple(volume_batch.shape[1:])
        mixed = model.input_buffer(mshape) if getattr(model, "volatile_io", False) else None      # the forward plan's own input tensor: no copy
        if mixed is None or mixed.dtype != volume_batch.dtype:
            mixed = torch.empty(mshape, dtype=volume_batch.dtype, device=volume_batch.device)
        BU.mix(uimg_a, img_a, img_mask, out=mixed[:lsub])      # net_input_unl, ACDC_BCP_train.py:372
        BU.mix(img_b, uimg_b, img_mask, out=mixed[lsub:])      # net_input_l,   :373
        model.drop_masks = cat_drops("s_unl", "s_l")
        out = model(mixed, groups=2)
        if side is not None:
            torch.cuda.current_stream(volume_batch.device).wait_stream(side)   # pseudo-labels are needed from here on
        if STEP_TOTAL:
            loss, unl_dice, unl_ce, l_dice, l_ce = BU.mix_loss_pair(out, (plab_a, lab_a, u_weight, 1.0), (lab_b, plab_b, 1.0, u_weight), loss_mask,
                                                                    flavour=H.LOSS_ACDC, total=True)     # loss: summed on the device, :381-384's order
            loss_ce, loss_dice = None, None      # (reported below from the detached terms)
        else:
            unl_dice, unl_ce, l_dice, l_ce = BU.mix_loss_pair(out, (plab_a, lab_a, u_weight, 1.0), (lab_b, plab_b, 1.0, u_weight), loss_mask,
                                                              flavour=H.LOSS_ACDC)
            loss_ce = unl_ce + l_ce
            loss_dice = unl_dice + l_dice
            loss = (loss_dice + loss_ce) / 2
        out_unl, out_l = out[:lsub], out[lsub:]
    else:
        net_input_unl = uimg_a * img_mask + img_a * (1 - img_mask)
        net_input_l = img_b * img_mask + uimg_b * (1 - img_mask)
        model.drop_masks = drops.get("s_unl")
        out_unl = model(net_input_unl)
        unl_dice, unl_ce = acdc_mix_loss(out_unl, plab_a, lab_a, loss_mask, u_weight=u_weight, unlab=True)
        model.drop_masks = drops.get("s_l")
        out_l = model(net_input_l)
        l_dice, l_ce = acdc_mix_loss(out_l, lab_b, plab_b, loss_mask, u_weight=u_weight)
        loss_ce = unl_ce + l_ce
        loss_dice = unl_dice + l_dice
        loss = (loss_dice + loss_ce) / 2
    if optimizer is None:              # gradient-only mode: the caller owns zero_grad / step / EMA
        _backward(model, loss)
    else:
        if not early_zero:
            optimizer.zero_grad()
        if dp is not None:
            dp.arm(model)              # one backward covers both student batches (grouped or not: `loss` sums their terms)
        _backward(model, loss)
        if dp is not None:
            dp.allreduce_grads(model, optimizer)
        optimizer.step()
        update_model_ema(model, ema_model, alpha)
    model.drop_masks = None
    ema_model.drop_masks = None
    if loss_ce is None:                # grouped: the two reported sums are side results, computed after the backward pass was enqueued
        loss_ce, loss_dice = unl_ce + l_ce, unl_dice + l_dice
    # (out_unl / out_l: the aliasing contract of la_self_train_step's outputs_l / outputs_u under volatile_io)
    return dict(loss=loss.detach(), loss_dice=loss_dice.detach(), loss_ce=loss_ce.detach(), plab_a=own_plabs[0], plab_b=own_plabs[1],
                out_unl=out_unl.detach(), out_l=out_l.detach())


def acdc_pre_train_step(model, optimizer, volume_batch, label_batch, box=None):
    """One ACDC pre-training iteration, ACDC_BCP_train.py:236-256: image a with a box of image b pasted in, trained against both
    label maps through mix_loss(u_weight=1.0, unlab=True); loss = (dice + ce) / 2"""
    sub_bs = volume_batch.shape[0] // 2
    img_a, img_b = volume_batch[:sub_bs], volume_batch[sub_bs:2 * sub_bs]
    lab_a, lab_b = label_batch[:sub_bs], label_batch[sub_bs:2 * sub_bs]
    if box is None:
        img_mask, loss_mask = generate_mask(img_a)
    else:
        sp = tuple(volume_batch.shape[2:])
        img_mask, loss_mask = BU.BoxMask(box, sp, None, False, volume_batch.device), BU.BoxMask(box, sp, sub_bs, False, volume_batch.device)
    net_input = img_a * img_mask + img_b * (1 - img_mask)
    out = model(net_input)
    loss_dice, loss_ce = acdc_mix_loss(out, lab_a, lab_b, loss_mask, u_weight=1.0, unlab=True)
    loss = (loss_dice + loss_ce) / 2
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return dict(loss=loss.detach(), loss_dice=loss_dice.detach(), loss_ce=loss_ce.detach())

