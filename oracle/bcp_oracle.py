"""oracle/bcp_oracle.py -- CPU restatement of the BCP self-training hot path.  TEST INFRASTRUCTURE.

This file is the *checker* for the HIP path, never the product: only tests/, bench.py's
`cpu_baseline` leg and __graft_entry__.smoke() may import it (see DESIGN.md "oracle").
It is written from SURVEY.md section 8a in plain functional torch-CPU fp32 (the path is
floating point), each function citing the reference lines it restates, and is pinned by the
golden vectors in tests/golden/ that oracle/make_golden.py captured by importing the reference
itself in the build container (tests/test_oracle_golden.py).

Parity status: pinned against the imported reference for every function below EXCEPT the
connected-component tie-break inside `largest_cc`: the reference calls skimage.measure.label
(scikit-image, unpinned, absent from the image), for which scipy.ndimage.label with the same
connectivity is the stand-in -- same component sets, raster-order numbering; "parity unpinned"
at that one third-party boundary.

All tensors here use the reference's logical layouts (NCDHW / NCHW, int64 labels).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------
# parameter dictionaries (reference state_dict keys and shapes)
# ----------------------------------------------------------------------------------------

VNET_FILTERS = 16


def _bn_keys(P, prefix, c, dims_track=True):
    P[prefix + ".weight"] = (c,)
    P[prefix + ".bias"] = (c,)
    P[prefix + ".running_mean"] = (c,)
    P[prefix + ".running_var"] = (c,)
    P[prefix + ".num_batches_tracked"] = ()


def _head_keys(P, n_sel):
    """The never-called contrastive heads (networks/VNet.py:250-278, networks/unet.py:216-247)."""
    for name, din in (("projection_head", 16), ("prediction_head", 32)):
        P[f"{name}.0.weight"] = (32, din)
        P[f"{name}.0.bias"] = (32,)
        _bn_keys(P, f"{name}.1", 32)
        P[f"{name}.3.weight"] = (32, 32)
        P[f"{name}.3.bias"] = (32,)
    for fam in ("contrastive_class_selector_", "contrastive_class_selector_memory"):
        for c in range(n_sel):
            P[f"{fam}{c}.0.weight"] = (32, 32)
            P[f"{fam}{c}.0.bias"] = (32,)
            _bn_keys(P, f"{fam}{c}.1", 32)
            P[f"{fam}{c}.3.weight"] = (1, 32)
            P[f"{fam}{c}.3.bias"] = (1,)


def vnet_layers(prefix_enc="encoder.", prefix_dec="decoder."):
    """Layer list of the LA V-Net (networks/VNet.py:145-239): (kind, key-prefix, cin, cout).

    kind: 'c3' 3x3x3 pad 1, 'dw' k2 s2 conv, 'up' k2 s2 transposed conv.  Each is followed by
    norm + ReLU; the list order is execution order."""
    nf = VNET_FILTERS
    L = []

    def block(name, n, cin, cout, pre):
        for i in range(n):
            L.append(("c3", f"{pre}{name}.conv.{3 * i}", cin if i == 0 else cout, cout))

    e, d = prefix_enc, prefix_dec
    block("block_one", 1, 1, nf, e)
    L.append(("dw", f"{e}block_one_dw.conv.0", nf, 2 * nf))
    block("block_two", 2, 2 * nf, 2 * nf, e)
    L.append(("dw", f"{e}block_two_dw.conv.0", 2 * nf, 4 * nf))
    block("block_three", 3, 4 * nf, 4 * nf, e)
    L.append(("dw", f"{e}block_three_dw.conv.0", 4 * nf, 8 * nf))
    block("block_four", 3, 8 * nf, 8 * nf, e)
    L.append(("dw", f"{e}block_four_dw.conv.0", 8 * nf, 16 * nf))
    block("block_five", 3, 16 * nf, 16 * nf, e)
    L.append(("up", f"{d}block_five_up.conv.0", 16 * nf, 8 * nf))
    block("block_six", 3, 8 * nf, 8 * nf, d)
    L.append(("up", f"{d}block_six_up.conv.0", 8 * nf, 4 * nf))
    block("block_seven", 3, 4 * nf, 4 * nf, d)
    L.append(("up", f"{d}block_seven_up.conv.0", 4 * nf, 2 * nf))
    block("block_eight", 2, 2 * nf, 2 * nf, d)
    L.append(("up", f"{d}block_eight_up.conv.0", 2 * nf, nf))
    return L


def vnet_param_shapes(n_classes=2, in_chns=1, variant="la"):
    """Ordered {key: shape} of the reference state_dict.

    variant 'la': networks/VNet.py VNet(normalization='batchnorm') -- 259 keys.
    variant 'pancreas': pancreas/Vnet.py VNet() (InstanceNorm3d affine=False, no buffers) -- 60 keys.
    """
    P = OrderedDict()
    la = variant == "la"
    enc, dec = ("encoder.", "decoder.") if la else ("", "")
    layers = vnet_layers(enc, dec)
    if in_chns != 1:
        k, p, _, co = layers[0]
        layers[0] = (k, p, in_chns, co)

    def emit(kind, pre, cin, cout):
        if kind == "up":
            P[pre + ".weight"] = (cin, cout, 2, 2, 2)
        elif kind == "dw":
            P[pre + ".weight"] = (cout, cin, 2, 2, 2)
        else:
            P[pre + ".weight"] = (cout, cin, 3, 3, 3)
        P[pre + ".bias"] = (cout,)
        if la:
            head, idx = pre.rsplit(".", 1)
            _bn_keys(P, f"{head}.{int(idx) + 1}", cout)

    if la:
        # registration order of the reference: encoder blocks, then decoder blocks
        for kind, pre, cin, cout in layers:
            emit(kind, pre, cin, cout)
            if pre.startswith("decoder.block_eight_up"):
                pass
        emit("c3", "decoder.block_nine.conv.0", VNET_FILTERS, VNET_FILTERS)
        P["decoder.out_conv.weight"] = (n_classes, VNET_FILTERS, 1, 1, 1)
        P["decoder.out_conv.bias"] = (n_classes,)
        _head_keys(P, 2)
    else:
        for kind, pre, cin, cout in layers:
            emit(kind, pre, cin, cout)
        emit("c3", "branchs.0.0.conv.0", VNET_FILTERS, VNET_FILTERS)
        P["branchs.0.1.weight"] = (n_classes, VNET_FILTERS, 1, 1, 1)
        P["branchs.0.1.bias"] = (n_classes,)
    return P


UNET_CH = [16, 32, 64, 128, 256]
UNET_DROP = [0.05, 0.1, 0.2, 0.3, 0.5]


def unet_param_shapes(n_classes=4, in_chns=1):
    """networks/unet.py UNet_2d state_dict (226 keys)."""
    P = OrderedDict()

    def convblock(pre, cin, cout):
        P[f"{pre}.conv_conv.0.weight"] = (cout, cin, 3, 3)
        P[f"{pre}.conv_conv.0.bias"] = (cout,)
        _bn_keys(P, f"{pre}.conv_conv.1", cout)
        P[f"{pre}.conv_conv.4.weight"] = (cout, cout, 3, 3)
        P[f"{pre}.conv_conv.4.bias"] = (cout,)
        _bn_keys(P, f"{pre}.conv_conv.5", cout)

    c = UNET_CH
    convblock("encoder.in_conv", in_chns, c[0])
    for i in range(1, 5):
        convblock(f"encoder.down{i}.maxpool_conv.1", c[i - 1], c[i])
    for i in range(1, 5):
        c1, c2 = c[5 - i], c[4 - i]
        P[f"decoder.up{i}.conv1x1.weight"] = (c2, c1, 1, 1)
        P[f"decoder.up{i}.conv1x1.bias"] = (c2,)
        convblock(f"decoder.up{i}.conv", 2 * c2, c2)
    P["decoder.out_conv.weight"] = (n_classes, c[0], 3, 3)
    P["decoder.out_conv.bias"] = (n_classes,)
    _head_keys(P, 4)
    return P


def init_params(shapes, seed=0, dtype=torch.float32, random_affine=False):
    """numpy-seeded (PCG64) synthetic weights scaled like torch's default init
    (uniform +-1/sqrt(fan_in)); BN gamma=1 beta=0 (or, with random_affine, gamma~U(0.5,1.5),
    beta~U(-0.2,0.2) so that parity tests exercise them), running_mean=0 running_var=1."""
    rng = np.random.default_rng(seed)
    P = OrderedDict()
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            P[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_mean"):
            P[k] = torch.zeros(shp, dtype=dtype)
        elif k.endswith("running_var"):
            P[k] = torch.ones(shp, dtype=dtype)
        elif len(shp) == 1 and (k.endswith(".weight")):
            P[k] = torch.ones(shp, dtype=dtype)  # BN gamma
            if random_affine:
                P[k] = torch.from_numpy(rng.uniform(0.5, 1.5, size=shp).astype(np.float32)).to(dtype)
        elif len(shp) == 1 and k.endswith(".bias") and (k[:-5] + ".running_mean") in shapes:
            P[k] = torch.zeros(shp, dtype=dtype)  # BN beta
            if random_affine:
                P[k] = torch.from_numpy(rng.uniform(-0.2, 0.2, size=shp).astype(np.float32)).to(dtype)
        else:
            wshape = shapes[k[:-5] + ".weight"] if len(shp) == 1 else shp
            fan_in = int(np.prod(wshape[1:]))
            b = 1.0 / math.sqrt(max(fan_in, 1))
            P[k] = torch.from_numpy(rng.uniform(-b, b, size=shp).astype(np.float32)).to(dtype)
    return P


def trainable_keys(shapes):
    return [k for k in shapes if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]


# ----------------------------------------------------------------------------------------
# networks
# ----------------------------------------------------------------------------------------

def _norm_act(y, P, bn_prefix, norm, train, momentum=0.1, eps=1e-5):
    """BatchNorm3d/2d (train: batch stats, running stats updated in place with UNBIASED var,
    networks/VNet.py:18-26) or InstanceNorm3d(affine=False) (pancreas/Vnet.py:93)."""
    if norm == "batchnorm":
        rm, rv = P[bn_prefix + ".running_mean"], P[bn_prefix + ".running_var"]
        out = F.batch_norm(y, rm, rv, P[bn_prefix + ".weight"], P[bn_prefix + ".bias"], train, momentum, eps)
        if train:
            P[bn_prefix + ".num_batches_tracked"] += 1
        return out
    if norm == "instancenorm":
        return F.instance_norm(y, eps=eps)
    raise ValueError(norm)


def _next(pre):
    head, idx = pre.rsplit(".", 1)
    return f"{head}.{int(idx) + 1}"


def vnet_forward(P, x, drop_masks=None, train=True, variant="la", has_dropout=True, act_masks=None):
    """LA V-Net forward (networks/VNet.py:167-186, 213-239, 286-290) or pancreas V-Net
    (pancreas/Vnet.py:137-194).  Returns logits [N,ncls,X,Y,Z].

    drop_masks: None (no dropout) or dict {'x5': [N,256] 0/1, 'x9': [N,16] 0/1} keep-masks of
    the two Dropout3d(p=0.5) sites (VNet.py:182-183, 236-237); kept channels are scaled by 2.
    act_masks: test hook -- one boolean tensor per norm layer, in execution order: ReLU(z) is evaluated as z * mask, i.e. the
    network is linearised on a GIVEN activation pattern (the one another fp32 implementation took), which removes the
    "pre-activation within an ulp of zero flips" noise from a gradient comparison without touching anything else.
    """
    la = variant == "la"
    norm = "batchnorm" if la else "instancenorm"
    enc, dec = ("encoder.", "decoder.") if la else ("", "")
    layers = vnet_layers(enc, dec)

    def run(kind, pre, h):
        w, b = P[pre + ".weight"], P[pre + ".bias"]
        if kind == "c3":
            y = F.conv3d(h, w, b, padding=1)
        elif kind == "dw":
            y = F.conv3d(h, w, b, stride=2)
        else:
            y = F.conv_transpose3d(h, w, b, stride=2)
        z = _norm_act(y, P, _next(pre), norm, train)
        if act_masks is not None:
            m = act_masks[len(used)].to(z.dtype)
            used.append(1)
            return z * m
        return F.relu(z)

    used = []
    feats = {}
    h = x
    skips = []
    i = 0
    # encoder: blocks one..five with dw after one..four
    for kind, pre, cin, cout in layers:
        name = pre.split("block_")[1].split(".")[0]
        if kind == "up":
            if name == "five_up" and la and has_dropout and drop_masks is not None:
                m = drop_masks["x5"].to(h.dtype).view(h.shape[0], -1, 1, 1, 1)
                h = h * m * 2.0
            h = run(kind, pre, h)
            h = h + skips.pop()          # skip add AFTER the up-block's ReLU (VNet.py:220-233)
        elif kind == "dw":
            skips.append(h)
            h = run(kind, pre, h)
        else:
            h = run(kind, pre, h)
    if la:
        h = run("c3", "decoder.block_nine.conv.0", h)
        if has_dropout and drop_masks is not None:
            m = drop_masks["x9"].to(h.dtype).view(h.shape[0], -1, 1, 1, 1)
            h = h * m * 2.0
        return F.conv3d(h, P["decoder.out_conv.weight"], P["decoder.out_conv.bias"])
    h = run("c3", "branchs.0.0.conv.0", h)
    return F.conv3d(h, P["branchs.0.1.weight"], P["branchs.0.1.bias"])


def unet_forward(P, x, drop_masks=None, train=True, act_masks=None, pool_idx=None):
    """UNet_2d forward (networks/unet.py:15-57, 80-86, 104-116, 254-257): logits [N,4,H,W].

    drop_masks: None or dict {'d0'..'d4': keep-mask tensors shaped like the activation they
    gate} for the five encoder nn.Dropout(p) sites (elementwise; kept values scaled 1/(1-p));
    decoder ConvBlocks use p=0.0."""

    used = []

    def lrelu(z):
        if act_masks is None:
            return F.leaky_relu(z, 0.01)
        m = act_masks[len(used)].to(z.dtype)      # act_masks: see vnet_forward (here: slope 1 where set, 0.01 elsewhere)
        used.append(1)
        return z * (m + 0.01 * (1.0 - m))

    def convblock(pre, h, dkey, p):
        y = F.conv2d(h, P[f"{pre}.conv_conv.0.weight"], P[f"{pre}.conv_conv.0.bias"], padding=1)
        y = lrelu(_norm_act(y, P, f"{pre}.conv_conv.1", "batchnorm", train))
        if drop_masks is not None and dkey is not None and p > 0:
            y = y * drop_masks[dkey].to(y.dtype) / (1.0 - p)
        y = F.conv2d(y, P[f"{pre}.conv_conv.4.weight"], P[f"{pre}.conv_conv.4.bias"], padding=1)
        return lrelu(_norm_act(y, P, f"{pre}.conv_conv.5", "batchnorm", train))

    def pool(h, i):
        if pool_idx is None:
            return F.max_pool2d(h, 2)
        idx = pool_idx[i]                          # test hook: the window winners ANOTHER implementation picked (flat H*W indices,
        return h.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)   # as F.max_pool2d(return_indices=True) numbers them)

    xs = [convblock("encoder.in_conv", x, "d0", UNET_DROP[0])]
    for i in range(1, 5):
        xs.append(convblock(f"encoder.down{i}.maxpool_conv.1", pool(xs[-1], i - 1), f"d{i}", UNET_DROP[i]))
    h = xs[4]
    for i in range(1, 5):
        h = F.conv2d(h, P[f"decoder.up{i}.conv1x1.weight"], P[f"decoder.up{i}.conv1x1.bias"])
        h = F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=True)
        h = torch.cat([xs[4 - i], h], dim=1)
        h = convblock(f"decoder.up{i}.conv", h, None, 0.0)
    return F.conv2d(h, P["decoder.out_conv.weight"], P["decoder.out_conv.bias"], padding=1)


# ----------------------------------------------------------------------------------------
# BCP ops
# ----------------------------------------------------------------------------------------

def box_la(rng_randint, img_shape=(112, 112, 80), mask_ratio=2 / 3):
    """context_mask box (utils/BCP_utils.py:18-28): three np.random.randint draws in w,h,z order;
    draw bounds are hard-coded 112/112/80 in the reference.  Returns (w,h,z,pw,ph,pz)."""
    px, py, pz = int(img_shape[0] * mask_ratio), int(img_shape[1] * mask_ratio), int(img_shape[2] * mask_ratio)
    w = rng_randint(0, 112 - px)
    h = rng_randint(0, 112 - py)
    z = rng_randint(0, 80 - pz)
    return (w, h, z, px, py, pz)


def box_acdc(rng_randint, img_shape=(256, 256)):
    """generate_mask (ACDC_BCP_train.py:131-140)."""
    px, py = int(img_shape[0] * 2 / 3), int(img_shape[1] * 2 / 3)
    w = rng_randint(0, img_shape[0] - px)
    h = rng_randint(0, img_shape[1] - py)
    return (w, h, px, py)


def box_pancreas(rng_randint, patch_size=64):
    """generate_mask (pancreas/pancreas_utils.py:187-200), volume hard-coded to 96^3."""
    w = rng_randint(0, 96 - patch_size)
    h = rng_randint(0, 96 - patch_size)
    z = rng_randint(0, 96 - patch_size)
    return (w, h, z, patch_size, patch_size, patch_size)


def box_to_mask(box, spatial, batch):
    """(mask int64 [spatial], loss_mask int64 [batch, spatial]); 1 outside the box, 0 inside."""
    mask = torch.ones(spatial, dtype=torch.int64)
    if len(spatial) == 3:
        w, h, z, pw, ph, pz = box
        mask[w:w + pw, h:h + ph, z:z + pz] = 0
    else:
        w, h, pw, ph = box
        mask[w:w + pw, h:h + ph] = 0
    return mask, mask.unsqueeze(0).repeat(batch, *([1] * len(spatial))).contiguous()


def mix(a, b, mask):
    """a*mask + b*(1-mask) (LA_BCP_train.py:248-251, ACDC_BCP_train.py:372-373)."""
    return a * mask + b * (1 - mask)


def get_cut_mask(out, thres=0.5):
    """softmax -> (p >= thres) -> channel 1 as int64 [N,X,Y,Z] (LA_BCP_train.py:57-60)."""
    probs = F.softmax(out, 1)
    return (probs >= thres).to(torch.int64)[:, 1].contiguous()


def get_acdc_argmax(out):
    """softmax -> argmax (first max wins) (ACDC_BCP_train.py:112-114)."""
    return torch.max(F.softmax(out, dim=1), dim=1)[1]


def _label(a, connectivity):
    from scipy import ndimage
    nd = a.ndim
    st = ndimage.generate_binary_structure(nd, connectivity if connectivity else nd)
    lab, _ = ndimage.label(a != 0, structure=st)
    return lab


def largest_cc(seg, connectivity=None):
    """LargestCC_pancreas (LA_BCP_train.py:65-77; pancreas_utils.py:284-296): per sample keep the
    largest connected component (ties -> lowest label id = first in raster order); empty ->
    pass-through.  Returns float32 [N,...] like the reference's torch.Tensor(list)."""
    out = []
    for n in range(seg.shape[0]):
        a = seg[n].cpu().numpy()
        lab = _label(a, connectivity)
        if lab.max() != 0:
            keep = lab == (np.argmax(np.bincount(lab.flat)[1:]) + 1)
        else:
            keep = a
        out.append(torch.from_numpy(np.ascontiguousarray(keep).astype(np.float32)))
    return torch.stack(out)


def largest_cc_acdc(seg):
    """get_ACDC_2DLargestCC (ACDC_BCP_train.py:89-109): per slice, per class c in 1..3, keep the
    largest 8-connected component of (seg==c), times c, summed."""
    out = []
    for n in range(seg.shape[0]):
        a = seg[n].cpu().numpy()
        acc = np.zeros(a.shape, dtype=np.float32)
        for c in range(1, 4):
            b = (a == c)
            lab = _label(b, None)
            if lab.max() != 0:
                keep = lab == (np.argmax(np.bincount(lab.flat)[1:]) + 1)
                acc += keep.astype(np.float32) * c
            else:
                acc += b.astype(np.float32)
        out.append(torch.from_numpy(acc))
    return torch.stack(out)


def mask_dice_loss(logits, target, mask=None, smooth=1e-5):
    """mask_DiceLoss.forward (utils/losses.py:47-77): per-(n,c) soft dice, mean over N*C."""
    N, C = logits.shape[0], logits.shape[1]
    p = F.softmax(logits.reshape(N, C, -1), dim=1)
    t = F.one_hot(target.reshape(N, -1).long(), C).permute(0, 2, 1).to(torch.float32)
    inter, union = p * t, p + t
    if mask is not None:
        m = mask.reshape(N, 1, -1)
        inter, union = inter * m, union * m
    inter, union = inter.sum(2), union.sum(2)
    return 1 - ((2 * inter + smooth) / (union + smooth)).mean()


def mix_loss_la(out, img_l, patch_l, mask, l_weight=1.0, u_weight=0.5, unlab=False):
    """mix_loss (utils/BCP_utils.py:58-69; pancreas/losses.py:129-141)."""
    img_l, patch_l = img_l.long(), patch_l.long()
    iw, pw = (u_weight, l_weight) if unlab else (l_weight, u_weight)
    pm = 1 - mask
    dice = mask_dice_loss(out, img_l, mask) * iw + mask_dice_loss(out, patch_l, pm) * pw
    ce = iw * (F.cross_entropy(out, img_l, reduction="none") * mask).sum() / (mask.sum() + 1e-16)
    ce = ce + pw * (F.cross_entropy(out, patch_l, reduction="none") * pm).sum() / (pm.sum() + 1e-16)
    return (dice + ce) / 2


def dice_loss_acdc(prob, target, mask, n_classes=4):
    """losses.DiceLoss.forward with mask (utils/losses.py:102-134): per class over the whole
    batch, squared denominators, smooth 1e-10, mean over classes."""
    loss = 0.0
    m = mask.to(torch.float32)
    for i in range(n_classes):
        s = prob[:, i]
        t = (target == i).to(torch.float32)
        inter = (s * t * m).sum()
        y = (t * t * m).sum()
        z = (s * s * m).sum()
        loss = loss + (1 - (2 * inter + 1e-10) / (z + y + 1e-10))
    return loss / n_classes


def mix_loss_acdc(out, img_l, patch_l, mask, l_weight=1.0, u_weight=0.5, unlab=False):
    """ACDC mix_loss (ACDC_BCP_train.py:167-179) -> (loss_dice, loss_ce)."""
    img_l, patch_l = img_l.long(), patch_l.long()
    soft = F.softmax(out, dim=1)
    iw, pw = (u_weight, l_weight) if unlab else (l_weight, u_weight)
    pm = 1 - mask
    dice = dice_loss_acdc(soft, img_l, mask) * iw + dice_loss_acdc(soft, patch_l, pm) * pw
    ce = iw * (F.cross_entropy(out, img_l, reduction="none") * mask).sum() / (mask.sum() + 1e-16)
    ce = ce + pw * (F.cross_entropy(out, patch_l, reduction="none") * pm).sum() / (pm.sum() + 1e-16)
    return dice, ce


def sup_loss_la(out, label):
    """LA / pancreas pre-train loss (LA_BCP_train.py:159-161): (mean CE + unmasked dice)/2."""
    return (F.cross_entropy(out, label.long()) + mask_dice_loss(out, label.long())) / 2


@torch.no_grad()
def ema_params(P_student, P_teacher, keys, alpha):
    """update_ema_variables (utils/BCP_utils.py:78-81): parameters only."""
    for k in keys:
        P_teacher[k].mul_(alpha).add_((1 - alpha) * P_student[k])


@torch.no_grad()
def ema_state_dict(P_student, P_teacher, alpha):
    """update_model_ema (ACDC_BCP_train.py:123-129): every state_dict entry, including BN
    buffers; int64 num_batches_tracked goes through float and is truncated on load."""
    for k in P_student:
        new = alpha * P_teacher[k] + (1 - alpha) * P_student[k]
        P_teacher[k].copy_(new.to(P_teacher[k].dtype))


@torch.no_grad()
def sgd_step(P, grads, bufs, keys, lr, momentum=0.9, weight_decay=1e-4):
    """torch.optim.SGD step (LA_BCP_train.py:218): g += wd*p; buf = m*buf + g (first step buf=g);
    p -= lr*buf.  keys with grad None are skipped."""
    for k in keys:
        g = grads.get(k)
        if g is None:
            continue
        g = g + weight_decay * P[k]
        if k not in bufs:
            bufs[k] = g.clone()
        else:
            bufs[k].mul_(momentum).add_(g)
        P[k].add_(bufs[k], alpha=-lr)


@torch.no_grad()
def adam_step(P, grads, state, keys, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (pancreas/dataloaders.py:182)."""
    for k in keys:
        g = grads.get(k)
        if g is None:
            continue
        st = state.setdefault(k, {"t": 0, "m": torch.zeros_like(g), "v": torch.zeros_like(g)})
        st["t"] += 1
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** st["t"], 1 - b2 ** st["t"]
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        P[k].addcdiv_(st["m"], denom, value=-lr / bc1)


def dice_metric(pred, gt):
    """medpy.metric.binary.dc formula: 2|P&G| / (|P|+|G|) (used for the 'Dice vs ref' gate)."""
    pred, gt = pred.bool(), gt.bool()
    inter = (pred & gt).sum().item()
    den = pred.sum().item() + gt.sum().item()
    return 2.0 * inter / den if den > 0 else 0.0


# ----------------------------------------------------------------------------------------
# whole self-training steps (the unit the benchmark counts)
# ----------------------------------------------------------------------------------------

def _with_grad(P, keys):
    Q = OrderedDict()
    for k, v in P.items():
        Q[k] = v.detach().clone().requires_grad_(True) if k in keys else v
    return Q


def la_self_train_step(Ps, Pt, volume, label, box, drops, sub_bs, u_weight=0.5, variant="la",
                       connectivity=None):
    """One LA self-training step up to the gradients (LA_BCP_train.py:235-257).

    volume [B,1,X,Y,Z] laid out lab_a|lab_b|unlab_a|unlab_b, label [B,X,Y,Z] int64.
    drops: dict with keys 't_a','t_b','s_l','s_u' -> drop_masks (or None).
    Returns dict(loss_l, loss_u, loss, grads{key: tensor}, plab_a, plab_b, out_l, out_u).
    Ps/Pt BN buffers are updated in place as in train() mode."""
    lb = 2 * sub_bs
    img_a, img_b = volume[:sub_bs], volume[sub_bs:lb]
    lab_a, lab_b = label[:sub_bs], label[sub_bs:lb]
    unimg_a, unimg_b = volume[lb:lb + sub_bs], volume[lb + sub_bs:]
    with torch.no_grad():
        ua = vnet_forward(Pt, unimg_a, drops.get("t_a"), True, variant)
        ub = vnet_forward(Pt, unimg_b, drops.get("t_b"), True, variant)
        plab_a = largest_cc(get_cut_mask(ua), connectivity)
        plab_b = largest_cc(get_cut_mask(ub), connectivity)
        img_mask, loss_mask = box_to_mask(box, tuple(volume.shape[2:]), sub_bs)
    if variant == "la":
        mixl = mix(img_a, unimg_a, img_mask)
        mixu = mix(unimg_b, img_b, img_mask)
    else:  # pancreas direction table (train_pancreas.py:155-156)
        mixl = mix(unimg_a, img_b, img_mask)
        mixu = mix(img_a, unimg_b, img_mask)
    keys = set(k for k in trainable_keys(Ps))
    Q = _with_grad(Ps, keys)
    out_l = vnet_forward(Q, mixl, drops.get("s_l"), True, variant)
    out_u = vnet_forward(Q, mixu, drops.get("s_u"), True, variant)
    if variant == "la":
        loss_l = mix_loss_la(out_l, lab_a, plab_a, loss_mask, u_weight=u_weight)
        loss_u = mix_loss_la(out_u, plab_b, lab_b, loss_mask, u_weight=u_weight, unlab=True)
    else:  # train_pancreas.py:160,164
        loss_l = mix_loss_la(out_l, plab_a.long(), lab_b, loss_mask, unlab=True)
        loss_u = mix_loss_la(out_u, lab_a, plab_b.long(), loss_mask)
    loss = loss_l + loss_u
    loss.backward()
    grads = {k: Q[k].grad for k in Q if k in keys and Q[k].grad is not None}
    # BN buffers were updated on Q's (shared) buffer tensors == Ps's tensors
    return dict(loss_l=loss_l.detach(), loss_u=loss_u.detach(), loss=loss.detach(), grads=grads,
                plab_a=plab_a, plab_b=plab_b, out_l=out_l.detach(), out_u=out_u.detach(),
                mixl=mixl, mixu=mixu)


def acdc_self_train_step(Ps, Pt, volume, label, box, drops, lsub, usub, u_weight=0.5):
    """One ACDC self-training step up to the gradients (ACDC_BCP_train.py:358-383)."""
    lbs = 2 * lsub
    img_a, img_b = volume[:lsub], volume[lsub:lbs]
    uimg_a, uimg_b = volume[lbs:lbs + usub], volume[lbs + usub:]
    lab_a, lab_b = label[:lsub], label[lsub:lbs]
    with torch.no_grad():
        pre_a = unet_forward(Pt, uimg_a, drops.get("t_a"), True)
        pre_b = unet_forward(Pt, uimg_b, drops.get("t_b"), True)
        plab_a = largest_cc_acdc(get_acdc_argmax(pre_a))
        plab_b = largest_cc_acdc(get_acdc_argmax(pre_b))
        img_mask, loss_mask = box_to_mask(box, tuple(volume.shape[2:]), lsub)
    in_unl = mix(uimg_a, img_a, img_mask)
    in_l = mix(img_b, uimg_b, img_mask)
    keys = set(trainable_keys(Ps))
    Q = _with_grad(Ps, keys)
    out_unl = unet_forward(Q, in_unl, drops.get("s_unl"), True)
    out_l = unet_forward(Q, in_l, drops.get("s_l"), True)
    unl_dice, unl_ce = mix_loss_acdc(out_unl, plab_a, lab_a, loss_mask, u_weight=u_weight, unlab=True)
    l_dice, l_ce = mix_loss_acdc(out_l, lab_b, plab_b, loss_mask, u_weight=u_weight)
    loss_ce, loss_dice = unl_ce + l_ce, unl_dice + l_dice
    loss = (loss_dice + loss_ce) / 2
    loss.backward()
    grads = {k: Q[k].grad for k in Q if k in keys and Q[k].grad is not None}
    return dict(loss=loss.detach(), loss_dice=loss_dice.detach(), loss_ce=loss_ce.detach(), grads=grads,
                plab_a=plab_a, plab_b=plab_b, out_unl=out_unl.detach(), out_l=out_l.detach())


# ----------------------------------------------------------------------------------------
# synthetic inputs: shared by tests and bench.py so both sides see equal data
# ----------------------------------------------------------------------------------------

def synth_la_batch(batch, shape=(112, 112, 80), seed=1337):
    """image ~N(0,1) f32 [B,1,X,Y,Z]; label int64 {0,1}: one ellipsoid (~8 % fg) + satellites."""
    rng = np.random.default_rng(seed)
    X, Y, Z = shape
    img = rng.standard_normal((batch, 1, X, Y, Z), dtype=np.float32)
    gx, gy, gz = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")
    lab = np.zeros((batch, X, Y, Z), dtype=np.int64)
    for b in range(batch):
        c = np.array([X, Y, Z]) * (0.5 + 0.1 * (rng.random(3) - 0.5))
        r = np.array([X, Y, Z]) * (0.27 + 0.04 * rng.random(3))
        m = ((gx - c[0]) / r[0]) ** 2 + ((gy - c[1]) / r[1]) ** 2 + ((gz - c[2]) / r[2]) ** 2 <= 1.0
        for _ in range(3):
            sc = rng.random(3) * np.array([X, Y, Z])
            sr = 2.0 + 2.0 * rng.random()
            m |= ((gx - sc[0]) ** 2 + (gy - sc[1]) ** 2 + (gz - sc[2]) ** 2) <= sr * sr
        lab[b] = m
        img[b, 0] += 1.5 * m  # make the label weakly visible in the image
    return torch.from_numpy(img), torch.from_numpy(lab)


def synth_acdc_batch(batch, shape=(256, 256), seed=1337):
    """image ~U[0,1) f32 [B,1,H,W]; label int64 {0..3}: three nested discs, random centre."""
    rng = np.random.default_rng(seed)
    H, W = shape
    img = rng.random((batch, 1, H, W), dtype=np.float32)
    gy, gx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    lab = np.zeros((batch, H, W), dtype=np.int64)
    for b in range(batch):
        cy, cx = H * (0.4 + 0.2 * rng.random()), W * (0.4 + 0.2 * rng.random())
        d2 = (gy - cy) ** 2 + (gx - cx) ** 2
        base = min(H, W)
        for c, rr in ((1, 0.30), (2, 0.20), (3, 0.10)):
            lab[b][d2 <= (rr * base) ** 2] = c
        img[b, 0] = 0.6 * img[b, 0] + 0.1 * lab[b]
    return torch.from_numpy(img), torch.from_numpy(lab)


# ------------------------------------------------------------------------------------------ validation (8f-1)
def sliding_window_la(P, image, stride_xy, stride_z, patch_size, variant="la"):
    """utils/test_3d_patch.py:82-141 test_single_case with the V-Net in eval() mode (running statistics, no dropout).
    image: numpy [W,H,D].  Returns (label_map int64 [W,H,D], score_map float32 [W,H,D]) -- the reference's
    score_map[0] (all its channels hold the class-1 probability, :131)."""
    import math
    image = np.asarray(image, dtype=np.float32)
    w, h, d = image.shape
    pads = []
    for size, p in zip((w, h, d), patch_size):
        tot = max(p - size, 0)
        pads.append((tot // 2, tot - tot // 2))                                    # :98-100
    add_pad = any(l or r for l, r in pads)
    if add_pad:
        image = np.pad(image, pads, mode="constant", constant_values=0)           # :101-102
    ww, hh, dd = image.shape
    sx = math.ceil((ww - patch_size[0]) / stride_xy) + 1                           # :105-107
    sy = math.ceil((hh - patch_size[1]) / stride_xy) + 1
    sz = math.ceil((dd - patch_size[2]) / stride_z) + 1
    score = np.zeros(image.shape, dtype=np.float32)
    cnt = np.zeros(image.shape, dtype=np.float32)
    for x in range(sx):
        xs = min(stride_xy * x, ww - patch_size[0])
        for y in range(sy):
            ys = min(stride_xy * y, hh - patch_size[1])
            for z in range(sz):
                zs = min(stride_z * z, dd - patch_size[2])
                patch = torch.from_numpy(image[xs:xs + patch_size[0], ys:ys + patch_size[1], zs:zs + patch_size[2]][None, None].copy())
                with torch.no_grad():
                    logits = vnet_forward(P, patch, None, False, variant, has_dropout=False)
                    prob = F.softmax(logits, dim=1)[0, 1].numpy()                  # :124-128
                sl = (slice(xs, xs + patch_size[0]), slice(ys, ys + patch_size[1]), slice(zs, zs + patch_size[2]))
                score[sl] += prob
                cnt[sl] += 1
    score = score / cnt                                                            # :134
    label = (score > 0.5).astype(np.int64)                                         # :135
    if add_pad:
        sl = tuple(slice(l, l + s) for (l, _), s in zip(pads, (w, h, d)))
        label, score = label[sl], score[sl]
    return label, score


def sliding_window_pancreas(P, image, stride_xy, stride_z, patch_size):
    """pancreas/test_util.py:88-148 test_single_case (TMI=0) with the IN-V-Net in eval() mode: BOTH softmax channels are
    accumulated, label = argmax over the averaged scores (first max wins: class 1 only where its score is strictly larger).
    image: numpy [W,H,D].  Returns (label_map int64 [W,H,D], score_map float32 [2,W,H,D])."""
    import math
    image = np.asarray(image, dtype=np.float32)
    w, h, d = image.shape
    pads = []
    for size, p in zip((w, h, d), patch_size):
        tot = max(p - size, 0)
        pads.append((tot // 2, tot - tot // 2))                                    # :108-110
    add_pad = any(l or r for l, r in pads)
    if add_pad:
        image = np.pad(image, pads, mode="constant", constant_values=0)           # :111-112
    ww, hh, dd = image.shape
    sx = math.ceil((ww - patch_size[0]) / stride_xy) + 1                           # :115-117
    sy = math.ceil((hh - patch_size[1]) / stride_xy) + 1
    sz = math.ceil((dd - patch_size[2]) / stride_z) + 1
    score = np.zeros((2,) + image.shape, dtype=np.float32)
    cnt = np.zeros(image.shape, dtype=np.float32)
    for x in range(sx):
        xs = min(stride_xy * x, ww - patch_size[0])
        for y in range(sy):
            ys = min(stride_xy * y, hh - patch_size[1])
            for z in range(sz):
                zs = min(stride_z * z, dd - patch_size[2])
                patch = torch.from_numpy(image[xs:xs + patch_size[0], ys:ys + patch_size[1], zs:zs + patch_size[2]][None, None].copy())
                with torch.no_grad():
                    logits = vnet_forward(P, patch, None, False, "pancreas", has_dropout=False)
                    prob = F.softmax(logits, dim=1)[0].numpy()                     # :134-137
                sl = (slice(None), slice(xs, xs + patch_size[0]), slice(ys, ys + patch_size[1]), slice(zs, zs + patch_size[2]))
                score[sl] += prob
                cnt[sl[1:]] += 1
    score = score / cnt[None]                                                      # :142
    label = np.argmax(score, axis=0)                                               # :143
    if add_pad:
        sl = tuple(slice(l, l + s) for (l, _), s in zip(pads, (w, h, d)))
        label, score = label[sl], score[(slice(None),) + sl]
    return label, score


def eval_params(seed, variant="la"):
    """random-init V-Net weights + NON-trivial running statistics (eval-mode BatchNorm must use them): the parameter set of
    tests/golden/sw_la.npz (oracle/make_golden_eval.py)"""
    P = init_params(vnet_param_shapes(variant=variant), seed=seed, random_affine=True)
    rng = np.random.default_rng(seed + 1)
    for k in P:
        if k.endswith("running_mean"):
            P[k] = torch.from_numpy(rng.normal(0.0, 0.2, tuple(P[k].shape)).astype(np.float32))
        elif k.endswith("running_var"):
            P[k] = torch.from_numpy(rng.uniform(0.5, 1.5, tuple(P[k].shape)).astype(np.float32))
    return P


def dice_binary(pred, gt):
    """medpy.metric.binary.dc: 2|A&B| / (|A| + |B|), 0.0 for two empty masks"""
    a, b = np.asarray(pred) != 0, np.asarray(gt) != 0
    den = int(a.sum()) + int(b.sum())
    return 2.0 * int((a & b).sum()) / den if den else 0.0


def la_rotflip_crop(image, label, output_size, randint):
    """RandomRotFlip + RandomCrop of the LA pipeline (dataloaders/dataset.py:52-59, 184-214) with an injectable randint
    (called in the reference's order: k, axis, w1, h1, d1).  numpy in, numpy out."""
    k = randint(0, 4)                                                               # :53
    image, label = np.rot90(image, k), np.rot90(label, k)                           # :54-55
    axis = randint(0, 2)                                                            # :56
    image, label = np.flip(image, axis=axis).copy(), np.flip(label, axis=axis).copy()
    P = output_size
    if label.shape[0] <= P[0] or label.shape[1] <= P[1] or label.shape[2] <= P[2]:  # :190-198
        pw = max((P[0] - label.shape[0]) // 2 + 3, 0)
        ph = max((P[1] - label.shape[1]) // 2 + 3, 0)
        pd = max((P[2] - label.shape[2]) // 2 + 3, 0)
        image = np.pad(image, [(pw, pw), (ph, ph), (pd, pd)], mode="constant", constant_values=0)
        label = np.pad(label, [(pw, pw), (ph, ph), (pd, pd)], mode="constant", constant_values=0)
    w, h, d = image.shape
    w1, h1, d1 = randint(0, w - P[0]), randint(0, h - P[1]), randint(0, d - P[2])   # :202-204
    sl = (slice(w1, w1 + P[0]), slice(h1, h1 + P[1]), slice(d1, d1 + P[2]))
    return image[sl], label[sl]


def pancreas_crop(samples, output_size, randint=None):
    """RandomCrop (randint given; called in the reference's order w1, h1, d1) / CenterCrop (randint None) of the pancreas pipeline
    (pancreas/dataloaders.py:22-91): pad ALL axes by (P - n) // 2 + 1 per side when any axis is <= the patch, then crop.
    numpy in, numpy out."""
    P, x = output_size, samples[0]
    if x.shape[0] <= P[0] or x.shape[1] <= P[1] or x.shape[2] <= P[2]:              # :34-40
        pads = [(max((P[i] - x.shape[i]) // 2 + 1, 0),) * 2 for i in range(3)]
        samples = [np.pad(s, pads, mode="constant", constant_values=0) for s in samples]
    w, h, d = samples[0].shape
    if randint is not None:                                                         # :43-45
        w1, h1, d1 = randint(0, w - P[0]), randint(0, h - P[1]), randint(0, d - P[2])
    else:                                                                           # :76-78
        w1, h1, d1 = int(round((w - P[0]) / 2.)), int(round((h - P[1]) / 2.)), int(round((d - P[2]) / 2.))
    return [s[w1:w1 + P[0], h1:h1 + P[1], d1:d1 + P[2]] for s in samples]


def _nearest_zoom(a, out_hw):
    """scipy.ndimage.zoom(a, (OH / H, OW / W), order=0) restated: source index = floor(o * (in - 1) / (out - 1) + 0.5)
    (scipy 1.15 NI_ZoomShift, grid_mode=False; pinned by tests/golden/aug_acdc.npz)"""
    H, W = a.shape
    OH, OW = out_hw
    px = np.floor(np.arange(OH, dtype=np.float64) * ((H - 1) / (OH - 1)) + 0.5).astype(np.int64)
    py = np.floor(np.arange(OW, dtype=np.float64) * ((W - 1) / (OW - 1)) + 0.5).astype(np.int64)
    return a[px[:, None], py[None, :]]


def rotate_affine(angle_deg, shape):
    """matrix and offset scipy.ndimage.rotate(reshape=False) hands to affine_transform: {m00, m01, m10, m11, off0, off1}"""
    from scipy import special        # cosdg / sindg (cephes, degree arguments) are what scipy.ndimage.rotate itself calls
    c, s = float(special.cosdg(angle_deg)), float(special.sindg(angle_deg))
    m = np.array([[c, s], [-s, c]])
    ctr = (np.asarray(shape, dtype=np.float64) - 1) / 2
    off = ctr - m @ ctr
    return [m[0, 0], m[0, 1], m[1, 0], m[1, 1], off[0], off[1]]


def _nearest_rotate(a, angle_deg):
    """scipy.ndimage.rotate(a, angle, order=0, reshape=False) restated (NI_GeometricTransform, mode='constant', cval=0):
    c = ((0 + x * m0) + y * m1) + offset per axis; outside [0, len - 1] -> 0; else a[floor(c + 0.5)]"""
    H, W = a.shape
    m00, m01, m10, m11, o0, o1 = rotate_affine(angle_deg, (H, W))
    x = np.arange(H, dtype=np.float64)[:, None]
    y = np.arange(W, dtype=np.float64)[None, :]
    c0 = ((0.0 + x * m00) + y * m01) + o0
    c1 = ((0.0 + x * m10) + y * m11) + o1
    ok = ~((c0 < 0) | (c0 > H - 1) | (c1 < 0) | (c1 > W - 1))
    p = np.floor(c0 + 0.5).astype(np.int64).clip(0, H - 1)
    q = np.floor(c1 + 0.5).astype(np.int64).clip(0, W - 1)
    return np.where(ok, a[p, q], np.zeros((), dtype=a.dtype))


def acdc_random_generator(image, label, output_size, rand, randint):
    """RandomGenerator.__call__ of the ACDC pipeline (dataloaders/dataset.py:69-88; helpers :52-66) with injectable random
    sources, called in the reference's order: rand() [python random.random], then either randint k, axis (rot_flip) or
    rand(), randint angle (rotate).  numpy in -> (float32 [1,OH,OW], uint8 [OH,OW])."""
    if rand() > 0.5:                                                            # :78-79
        k = randint(0, 4)
        image, label = np.rot90(image, k), np.rot90(label, k)
        axis = randint(0, 2)
        image, label = np.flip(image, axis=axis).copy(), np.flip(label, axis=axis).copy()
    elif rand() > 0.5:                                                          # :80-81
        angle = randint(-20, 20)
        image, label = _nearest_rotate(image, angle), _nearest_rotate(label, angle)
    image = _nearest_zoom(image, output_size)                                   # :82-84
    label = _nearest_zoom(label, output_size)
    return image.astype(np.float32)[None], label.astype(np.uint8)               # :85-86
