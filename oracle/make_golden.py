"""oracle/make_golden.py -- capture golden vectors by IMPORTING THE REFERENCE (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
Reads /root/reference/code (read-only), writes tests/golden/*.npz|json.  Nothing of the
reference's source travels: the fixtures hold inputs/outputs only.  On the GPU box this
script is never run (no /root/reference there).

What is captured: state_dict keys/shapes of the three networks; forward +
backward of the reference networks on numpy-seeded weights/inputs with injected dropout masks
(tiny shapes stored fully; config shapes as fp64 checksums + strided samples); every BCP op
(box mask draws, mix, pseudo-label, largest-CC with the scipy stand-in for skimage, both
mix_loss flavours, supervised loss, both EMA flavours, SGD); and a 3-step LA / ACDC
self-training trajectory driven by the reference's own functions.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/code"
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.set_num_threads(8)


# ------------------------------------------------------------------ stubs for absent third-party modules
class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, n):
        return _Any()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__file__ = "<stub:%s>" % name

    def _ga(n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Any

    m.__getattr__ = _ga  # type: ignore
    sys.modules[name] = m
    return m


def _sk_label(a, connectivity=None, **kw):
    from scipy import ndimage
    a = np.asarray(a)
    st = ndimage.generate_binary_structure(a.ndim, connectivity if connectivity else a.ndim)
    return ndimage.label(a != 0, structure=st)[0]


for name in ["turtle", "tensorboardX", "medpy", "medpy.metric", "torchvision", "torchvision.transforms",
             "torchvision.utils", "h5py", "nibabel", "cv2", "imageio", "skimage", "skimage.transform",
             "skimage.segmentation", "SimpleITK", "nrrd", "itertools_stub"]:
    _stub(name)
_stub("skimage.measure", label=_sk_label)
sys.modules["skimage"].measure = sys.modules["skimage.measure"]
sys.modules["medpy"].metric = sys.modules["medpy.metric"]
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]

# no GPU here: the reference calls .cuda() unconditionally
torch.Tensor.cuda = lambda self, *a, **k: self
nn.Module.cuda = lambda self, *a, **k: self

sys.path.insert(0, REF)
sys.argv = ["x"]
from networks.net_factory import net_factory, BCP_net  # noqa: E402
from utils import losses as ref_losses  # noqa: E402
from utils import BCP_utils as ref_bcp  # noqa: E402
import LA_BCP_train as ref_la  # noqa: E402
import ACDC_BCP_train as ref_acdc  # noqa: E402

sys.path.insert(0, os.path.join(REF, "pancreas"))
import importlib.util  # noqa: E402


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_pvnet = _load(os.path.join(REF, "pancreas", "Vnet.py"), "ref_pancreas_vnet")
ref_plosses = _load(os.path.join(REF, "pancreas", "losses.py"), "ref_pancreas_losses")
sys.modules["statistic"] = _stub("statistic")
ref_putils = _load(os.path.join(REF, "pancreas", "pancreas_utils.py"), "ref_pancreas_utils")

import bcp_oracle as O  # noqa: E402


class InjectDrop3d(nn.Module):
    """Replaces nn.Dropout3d(p=0.5) in the reference net: keep-mask [N,C] injected."""

    def __init__(self):
        super().__init__()
        self.mask = None

    def forward(self, x):
        if self.mask is None:
            return x
        return x * self.mask.view(x.shape[0], -1, 1, 1, 1).to(x.dtype) * 2.0


class InjectDrop2d(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p
        self.mask = None

    def forward(self, x):
        if self.mask is None or self.p == 0:
            return x
        return x * self.mask.to(x.dtype) / (1.0 - self.p)


def ref_vnet_la(P):
    net = net_factory("VNet", in_chns=1, class_num=2, mode="train")
    net.load_state_dict(P, strict=True)
    net.encoder.dropout = InjectDrop3d()
    net.decoder.dropout = InjectDrop3d()
    # the MaxPool3d(3,2) on features[4] (VNet.py:289) feeds only the second, dead return value and
    # needs >=3 voxels per dim at the deepest level; drop it so tiny shapes run.
    net.pool = nn.Identity()
    net.train()
    return net


def set_drop_la(net, dm):
    net.encoder.dropout.mask = None if dm is None else dm["x5"]
    net.decoder.dropout.mask = None if dm is None else dm["x9"]


def ref_unet(P):
    net = BCP_net(in_chns=1, class_num=4)
    net.load_state_dict(P, strict=True)
    blocks = [net.encoder.in_conv] + [getattr(net.encoder, f"down{i}").maxpool_conv[1] for i in range(1, 5)]
    for b, p in zip(blocks, O.UNET_DROP):
        b.conv_conv[3] = InjectDrop2d(p)
    net.train()
    return net, blocks


def set_drop_unet(blocks, dm):
    for i, b in enumerate(blocks):
        b.conv_conv[3].mask = None if dm is None else dm[f"d{i}"]


def stats(t):
    a = t.detach().double().reshape(-1)
    return [float(a.sum()), float(a.abs().sum()), float((a * a).sum().sqrt())]


def sample(t, n=2048):
    a = t.detach().reshape(-1)
    idx = torch.linspace(0, a.numel() - 1, min(n, a.numel())).long()
    return a[idx].numpy().copy()


def clone_params(P):
    return {k: v.clone() for k, v in P.items()}


def la_drop_masks(rng, n):
    return {"x5": torch.from_numpy((rng.random((n, 256)) < 0.5).astype(np.float32)),
            "x9": torch.from_numpy((rng.random((n, 16)) < 0.5).astype(np.float32))}


def unet_drop_masks(rng, n, hw):
    dm = {}
    h, w = hw
    for i, (c, p) in enumerate(zip(O.UNET_CH, O.UNET_DROP)):
        dm[f"d{i}"] = torch.from_numpy((rng.random((n, c, h >> i, w >> i)) >= p).astype(np.float32))
    return dm


def main():
    os.makedirs(OUT, exist_ok=True)
    meta = {}

    # ------------------------------------------------------------ G1: state_dict keys / shapes
    net = net_factory("VNet", in_chns=1, class_num=2, mode="train")
    sd = net.state_dict()
    meta["vnet_la_keys"] = [[k, list(v.shape)] for k, v in sd.items()]
    meta["vnet_la_param_names"] = [n for n, _ in net.named_parameters()]
    net2 = BCP_net(in_chns=1, class_num=4)
    meta["unet_keys"] = [[k, list(v.shape)] for k, v in net2.state_dict().items()]
    meta["unet_param_names"] = [n for n, _ in net2.named_parameters()]
    net3 = ref_pvnet.VNet()
    meta["vnet_pancreas_keys"] = [[k, list(v.shape)] for k, v in net3.state_dict().items()]
    meta["vnet_pancreas_param_names"] = [n for n, _ in net3.named_parameters()]
    # defaults of norm layers the oracle must match
    meta["bn_eps"] = net.encoder.block_one.conv[1].eps
    meta["bn_momentum"] = net.encoder.block_one.conv[1].momentum
    meta["in_eps"] = net3.block_one.conv[1].eps
    meta["in_affine"] = bool(net3.block_one.conv[1].affine)
    meta["in_track"] = bool(net3.block_one.conv[1].track_running_stats)

    # ------------------------------------------------------------ G2: tiny LA V-Net fwd+bwd, stored fully
    shapes = O.vnet_param_shapes()
    P0 = O.init_params(shapes, seed=11, random_affine=True)
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((2, 1, 32, 32, 16), dtype=np.float32))
    tgt = torch.from_numpy(rng.integers(0, 2, (2, 32, 32, 16)))
    dm = la_drop_masks(rng, 2)
    net = ref_vnet_la(clone_params(P0))
    set_drop_la(net, dm)
    out, _ = net(x)
    loss = ref_bcp.sup_loss(out, tgt)
    loss.backward()
    g = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    sd = net.state_dict()
    np.savez_compressed(
        os.path.join(OUT, "vnet_la_tiny.npz"),
        x=x.numpy(), tgt=tgt.numpy(), drop_x5=dm["x5"].numpy(), drop_x9=dm["x9"].numpy(),
        logits=out.detach().numpy(), loss=np.float64(loss.item()),
        grad_names=np.array(list(g.keys())),
        grad_stats=np.array([stats(v) for v in g.values()]),
        grad_block_one_w=g["encoder.block_one.conv.0.weight"].numpy(),
        grad_block_nine_w=g["decoder.block_nine.conv.0.weight"].numpy(),
        grad_five_up_w=g["decoder.block_five_up.conv.0.weight"].numpy(),
        grad_one_dw_w=g["encoder.block_one_dw.conv.0.weight"].numpy(),
        grad_out_conv_w=g["decoder.out_conv.weight"].numpy(),
        grad_bn1_w=g["encoder.block_one.conv.1.weight"].numpy(),
        grad_bn1_b=g["encoder.block_one.conv.1.bias"].numpy(),
        rm_block_one=sd["encoder.block_one.conv.1.running_mean"].numpy(),
        rv_block_one=sd["encoder.block_one.conv.1.running_var"].numpy(),
        rm_block_nine=sd["decoder.block_nine.conv.1.running_mean"].numpy(),
        rv_block_nine=sd["decoder.block_nine.conv.1.running_var"].numpy(),
        nbt=np.int64(sd["encoder.block_one.conv.1.num_batches_tracked"].item()),
    )
    meta["vnet_la_tiny"] = {"param_seed": 11, "random_affine": True, "n_grads": len(g)}

    # ------------------------------------------------------------ G3: config-shape LA V-Net fwd+bwd (checksums)
    P1 = O.init_params(shapes, seed=1337, random_affine=True)
    xb, lb = O.synth_la_batch(1, seed=1337)
    rng = np.random.default_rng(6)
    dm = la_drop_masks(rng, 1)
    net = ref_vnet_la(clone_params(P1))
    set_drop_la(net, dm)
    out, _ = net(xb)
    loss = ref_bcp.sup_loss(out, lb)
    loss.backward()
    g = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    np.savez_compressed(
        os.path.join(OUT, "vnet_la_full.npz"),
        drop_x5=dm["x5"].numpy(), drop_x9=dm["x9"].numpy(),
        logits_stats=np.array(stats(out)), logits_sample=sample(out, 4096), loss=np.float64(loss.item()),
        grad_names=np.array(list(g.keys())), grad_stats=np.array([stats(v) for v in g.values()]),
        x_stats=np.array(stats(xb)), lab_sum=np.int64(lb.sum().item()),
    )
    meta["vnet_la_full"] = {"param_seed": 1337, "data_seed": 1337, "drop_seed": 6}

    # ------------------------------------------------------------ G4: BCP ops
    ops = {}
    np.random.seed(1337)
    boxes = []
    for _ in range(4):
        img = torch.zeros(2, 1, 112, 112, 80)
        m, lm = ref_bcp.context_mask(img, 2 / 3)
        zero = (m == 0).nonzero()
        lo, hi = zero.min(0)[0], zero.max(0)[0]
        boxes.append([int(lo[0]), int(lo[1]), int(lo[2]), int(hi[0] - lo[0] + 1), int(hi[1] - lo[1] + 1), int(hi[2] - lo[2] + 1)])
        assert int(m.sum()) == 112 * 112 * 80 - boxes[-1][3] * boxes[-1][4] * boxes[-1][5]
    ops["la_boxes_seed1337"] = boxes
    ops["la_mask_sum"] = int(m.sum())
    np.random.seed(1337)
    boxes = []
    for _ in range(4):
        m, lm = ref_acdc.generate_mask(torch.zeros(2, 1, 256, 256))
        zero = (m == 0).nonzero()
        lo, hi = zero.min(0)[0], zero.max(0)[0]
        boxes.append([int(lo[0]), int(lo[1]), int(hi[0] - lo[0] + 1), int(hi[1] - lo[1] + 1)])
    ops["acdc_boxes_seed1337"] = boxes
    np.random.seed(2020)
    boxes = []
    for _ in range(4):
        m, lm = ref_putils.generate_mask(torch.zeros(1, 1, 96, 96, 96), 64)
        zero = (m == 0).nonzero()
        lo, hi = zero.min(0)[0], zero.max(0)[0]
        boxes.append([int(lo[0]), int(lo[1]), int(lo[2]), 64, 64, 64])
    ops["pancreas_boxes_seed2020"] = boxes

    # mix_loss LA (SURVEY 8c smoke values re-captured)
    rng = np.random.default_rng(0)
    logits = torch.from_numpy(rng.standard_normal((2, 2, 16, 16, 8), dtype=np.float32)).requires_grad_(True)
    a = torch.from_numpy(rng.integers(0, 2, (2, 16, 16, 8)))
    b = torch.from_numpy(rng.integers(0, 2, (2, 16, 16, 8)))
    mask = torch.ones(2, 16, 16, 8, dtype=torch.int64)
    mask[:, 2:12, 3:13, 1:6] = 0
    l1 = ref_bcp.mix_loss(logits, a, b, mask, u_weight=0.5)
    l1.backward()
    g1 = logits.grad.clone()
    logits.grad = None
    l2 = ref_bcp.mix_loss(logits, a, b, mask, u_weight=0.5, unlab=True)
    l2.backward()
    g2 = logits.grad.clone()
    logits.grad = None
    l3 = ref_bcp.sup_loss(logits, a)
    l3.backward()
    g3 = logits.grad.clone()
    logits.grad = None
    lp = ref_plosses.mix_loss(logits, a, b, mask, unlab=True)
    ce = F.cross_entropy(logits, a)
    dice = ref_losses.mask_DiceLoss(2)(logits, a)
    np.savez_compressed(os.path.join(OUT, "mixloss_la.npz"), logits=logits.detach().numpy(), a=a.numpy(), b=b.numpy(),
                        mask=mask.numpy(), l1=np.float64(l1.item()), l2=np.float64(l2.item()), l3=np.float64(l3.item()),
                        lp=np.float64(lp.item()), g1=g1.numpy(), g2=g2.numpy(), g3=g3.numpy(),
                        ce=np.float64(ce.item()), dice=np.float64(dice.item()))
    ops["mixloss_la"] = {"l1": l1.item(), "l2": l2.item(), "sum_abs_g1": float(g1.abs().sum())}

    # mix_loss ACDC
    rng = np.random.default_rng(1)
    logits = torch.from_numpy(rng.standard_normal((2, 4, 32, 32), dtype=np.float32)).requires_grad_(True)
    a = torch.from_numpy(rng.integers(0, 4, (2, 32, 32)))
    b = torch.from_numpy(rng.integers(0, 4, (2, 32, 32)))
    mask = torch.ones(2, 32, 32, dtype=torch.int64)
    mask[:, 4:25, 6:27] = 0
    d1, c1 = ref_acdc.mix_loss(logits, a, b, mask, u_weight=0.5, unlab=True)
    ((d1 + c1) / 2).backward()
    g1 = logits.grad.clone()
    logits.grad = None
    d2, c2 = ref_acdc.mix_loss(logits, a, b, mask, u_weight=0.5)
    ((d2 + c2) / 2).backward()
    g2 = logits.grad.clone()
    logits.grad = None
    np.savez_compressed(os.path.join(OUT, "mixloss_acdc.npz"), logits=logits.detach().numpy(), a=a.numpy(), b=b.numpy(),
                        mask=mask.numpy(), d1=np.float64(d1.item()), c1=np.float64(c1.item()), d2=np.float64(d2.item()),
                        c2=np.float64(c2.item()), g1=g1.numpy(), g2=g2.numpy())
    ops["mixloss_acdc"] = {"d1": d1.item(), "c1": c1.item(), "d2": d2.item(), "c2": c2.item()}

    # pseudo-label + largest CC (LA 26-conn; pancreas 18-conn; ACDC 8-conn per class)
    rng = np.random.default_rng(2)
    lo = rng.standard_normal((2, 2, 24, 24, 16), dtype=np.float32)
    # smooth to obtain blobs, plus exact ties (p == 0.5 -> 1)
    lo_t = F.avg_pool3d(torch.from_numpy(lo), 5, 1, 2) * 6
    lo_t[0, :, 0, 0, :4] = 0.25  # equal logits: tie
    cut = ref_la.get_cut_mask(lo_t, nms=0)
    cc26 = ref_la.get_cut_mask(lo_t, nms=1)
    cc18 = ref_putils.get_cut_mask(lo_t, nms=True, connect_mode=2)
    cc6 = ref_putils.get_cut_mask(lo_t, nms=True, connect_mode=1)
    lo2 = F.avg_pool2d(torch.from_numpy(rng.standard_normal((3, 4, 48, 48), dtype=np.float32)), 5, 1, 2) * 8
    am = ref_acdc.get_ACDC_masks(lo2, nms=0)
    amcc = ref_acdc.get_ACDC_masks(lo2, nms=1)
    np.savez_compressed(os.path.join(OUT, "plabel_cc.npz"), logits3d=lo_t.numpy(), cut=cut.numpy().astype(np.uint8),
                        cc26=cc26.numpy().astype(np.uint8), cc18=cc18.numpy().astype(np.uint8),
                        cc6=cc6.numpy().astype(np.uint8), logits2d=lo2.numpy(), argmax=am.numpy().astype(np.uint8),
                        argmax_cc=amcc.numpy().astype(np.uint8))
    ops["cc_dtype"] = str(cc26.dtype)
    ops["cc_counts"] = [int(cut.sum()), int(cc26.sum()), int(cc18.sum()), int(cc6.sum())]

    # EMA flavours + SGD on a small net state (use the tiny V-Net / U-Net)
    netA = ref_vnet_la(clone_params(P0))
    netB = ref_vnet_la(clone_params(O.init_params(shapes, seed=12, random_affine=True)))
    ref_bcp.update_ema_variables(netA, netB, 0.99)
    sdB = netB.state_dict()
    ops["ema_la"] = {k: stats(sdB[k]) for k in ["encoder.block_one.conv.0.weight", "decoder.block_nine.conv.1.weight",
                                                 "encoder.block_one.conv.1.running_mean"]}
    ushapes = O.unet_param_shapes()
    U0 = O.init_params(ushapes, seed=21, random_affine=True)
    U1 = O.init_params(ushapes, seed=22, random_affine=True)
    for k in U1:
        if k.endswith("running_mean"):
            U1[k] = U1[k] + 0.25
        if k.endswith("num_batches_tracked"):
            U0[k] = U0[k] + 250
            U1[k] = U1[k] + 3
    uA, _ = ref_unet(clone_params(U0))
    uB, _ = ref_unet(clone_params(U1))
    ref_acdc.update_model_ema(uA, uB, 0.99)
    sdB = uB.state_dict()
    ops["ema_acdc"] = {k: stats(sdB[k]) for k in ["encoder.in_conv.conv_conv.0.weight", "encoder.in_conv.conv_conv.1.running_mean",
                                                   "encoder.in_conv.conv_conv.1.num_batches_tracked", "decoder.out_conv.bias"]}
    meta["ops"] = ops

    # ------------------------------------------------------------ G5: U-Net fwd+bwd tiny (stored) and config (checksums)
    rng = np.random.default_rng(7)
    x = torch.from_numpy(rng.random((2, 1, 64, 64), dtype=np.float32))
    tgt = torch.from_numpy(rng.integers(0, 4, (2, 64, 64)))
    dm = unet_drop_masks(rng, 2, (64, 64))
    unet, blocks = ref_unet(clone_params(U0))
    set_drop_unet(blocks, dm)
    out = unet(x)
    m1 = torch.ones(2, 64, 64, dtype=torch.int64)
    m1[:, 10:52, 8:50] = 0
    dl, cl = ref_acdc.mix_loss(out, tgt, (tgt + 1) % 4, m1, u_weight=0.5)
    loss = (dl + cl) / 2
    loss.backward()
    g = {n: p.grad for n, p in unet.named_parameters() if p.grad is not None}
    sd = unet.state_dict()
    np.savez_compressed(
        os.path.join(OUT, "unet_tiny.npz"), x=x.numpy(), tgt=tgt.numpy(),
        **{f"drop_d{i}": np.packbits(dm[f"d{i}"].numpy().astype(np.uint8)) for i in range(5)},
        logits=out.detach().numpy(), loss=np.float64(loss.item()),
        grad_names=np.array(list(g.keys())), grad_stats=np.array([stats(v) for v in g.values()]),
        grad_in_conv_w=g["encoder.in_conv.conv_conv.0.weight"].numpy(),
        grad_up4_1x1_w=g["decoder.up4.conv1x1.weight"].numpy(),
        grad_out_conv_w=g["decoder.out_conv.weight"].numpy(),
        rm_in=sd["encoder.in_conv.conv_conv.1.running_mean"].numpy(), rv_in=sd["encoder.in_conv.conv_conv.1.running_var"].numpy(),
    )
    meta["unet_tiny"] = {"param_seed": 21, "mask_box": [10, 8, 42, 42], "n_grads": len(g)}

    xb, lb = O.synth_acdc_batch(6, seed=1337)
    rng = np.random.default_rng(8)
    dm = unet_drop_masks(rng, 6, (256, 256))
    U2 = O.init_params(ushapes, seed=1337, random_affine=True)
    unet, blocks = ref_unet(clone_params(U2))
    set_drop_unet(blocks, dm)
    out = unet(xb)
    m1 = torch.ones(6, 256, 256, dtype=torch.int64)
    m1[:, 40:210, 30:200] = 0
    dl, cl = ref_acdc.mix_loss(out, lb, (lb + 1) % 4, m1, u_weight=0.5, unlab=True)
    loss = (dl + cl) / 2
    loss.backward()
    g = {n: p.grad for n, p in unet.named_parameters() if p.grad is not None}
    np.savez_compressed(os.path.join(OUT, "unet_full.npz"), logits_stats=np.array(stats(out)), logits_sample=sample(out, 4096),
                        dice=np.float64(dl.item()), ce=np.float64(cl.item()),
                        grad_names=np.array(list(g.keys())), grad_stats=np.array([stats(v) for v in g.values()]))
    meta["unet_full"] = {"param_seed": 1337, "data_seed": 1337, "drop_seed": 8, "mask_box": [40, 30, 170, 170]}

    # ------------------------------------------------------------ G6: pancreas V-Net tiny
    pshapes = O.vnet_param_shapes(variant="pancreas")
    PP = O.init_params(pshapes, seed=31)
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.standard_normal((1, 1, 32, 32, 32), dtype=np.float32))
    tgt = torch.from_numpy(rng.integers(0, 2, (1, 32, 32, 32)))
    pnet = ref_pvnet.VNet()
    pnet.load_state_dict(clone_params(PP), strict=True)
    pnet.train()
    out = pnet(x)[0]
    loss = (F.cross_entropy(out, tgt) + ref_plosses.DiceLoss(2)(out, tgt)) / 2
    loss.backward()
    g = {n: p.grad for n, p in pnet.named_parameters() if p.grad is not None}
    np.savez_compressed(os.path.join(OUT, "vnet_pancreas_tiny.npz"), x=x.numpy(), tgt=tgt.numpy(), logits=out.detach().numpy(),
                        loss=np.float64(loss.item()), grad_names=np.array(list(g.keys())),
                        grad_stats=np.array([stats(v) for v in g.values()]),
                        grad_block_one_w=g["block_one.conv.0.weight"].numpy(),
                        grad_head_w=g["branchs.0.1.weight"].numpy())
    meta["vnet_pancreas_tiny"] = {"param_seed": 31}

    # ------------------------------------------------------------ G7: 3-step LA self-training trajectory, reference functions
    # loop body = LA_BCP_train.py:235-270 driven with the reference's own callables on synthetic data
    P0s = O.init_params(shapes, seed=41, random_affine=True)
    model = ref_vnet_la(clone_params(P0s))
    ema = ref_vnet_la(clone_params(P0s))
    for p in ema.parameters():
        p.detach_()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=0.0001)
    traj = []
    rngd = np.random.default_rng(42)
    np.random.seed(1337)
    vol, lab = O.synth_la_batch(4, shape=(32, 32, 16), seed=77)
    # context_mask draws with bounds 112/80 regardless of the image: emulate with explicit small boxes instead
    boxes, drops_all = [], []
    sub_bs = 1
    for it in range(3):
        img_a, img_b = vol[:1], vol[1:2]
        lab_a, lab_b = lab[:1], lab[1:2]
        unimg_a, unimg_b = vol[2:3], vol[3:4]
        d = {k: la_drop_masks(rngd, 1) for k in ("t_a", "t_b", "s_l", "s_u")}
        drops_all.append(d)
        with torch.no_grad():
            set_drop_la(ema, d["t_a"])
            ua, _ = ema(unimg_a)
            set_drop_la(ema, d["t_b"])
            ub, _ = ema(unimg_b)
            plab_a = ref_la.get_cut_mask(ua, nms=1)
            plab_b = ref_la.get_cut_mask(ub, nms=1)
            w, h, z = int(rngd.integers(0, 32 - 21)), int(rngd.integers(0, 32 - 21)), int(rngd.integers(0, 16 - 10))
            box = (w, h, z, 21, 21, 10)
            boxes.append(box)
            img_mask, loss_mask = O.box_to_mask(box, (32, 32, 16), 1)
        mixl_img = img_a * img_mask + unimg_a * (1 - img_mask)
        mixu_img = unimg_b * img_mask + img_b * (1 - img_mask)
        set_drop_la(model, d["s_l"])
        outputs_l, _ = model(mixl_img)
        set_drop_la(model, d["s_u"])
        outputs_u, _ = model(mixu_img)
        loss_l = ref_bcp.mix_loss(outputs_l, lab_a, plab_a, loss_mask, u_weight=0.5)
        loss_u = ref_bcp.mix_loss(outputs_u, plab_b, lab_b, loss_mask, u_weight=0.5, unlab=True)
        loss = loss_l + loss_u
        opt.zero_grad()
        loss.backward()
        opt.step()
        ref_bcp.update_ema_variables(model, ema, 0.99)
        traj.append([loss.item(), loss_l.item(), loss_u.item(), float(plab_a.sum()), float(plab_b.sum())])
    sdm, sde = model.state_dict(), ema.state_dict()
    np.savez_compressed(
        os.path.join(OUT, "la_traj.npz"), traj=np.array(traj), boxes=np.array(boxes),
        drops=np.array([[np.concatenate([d[k]["x5"].numpy().ravel(), d[k]["x9"].numpy().ravel()]) for k in ("t_a", "t_b", "s_l", "s_u")]
                        for d in drops_all]),
        final_w_stats=np.array([stats(sdm[k]) for k in meta["vnet_la_param_names"][:60]]),
        final_ema_stats=np.array([stats(sde[k]) for k in meta["vnet_la_param_names"][:60]]),
        final_rm=sdm["decoder.block_nine.conv.1.running_mean"].numpy(),
        final_ema_rm=sde["decoder.block_nine.conv.1.running_mean"].numpy(),
    )
    meta["la_traj"] = {"param_seed": 41, "data_seed": 77, "shape": [32, 32, 16], "steps": 3}

    # ------------------------------------------------------------ G8: 2-step ACDC trajectory
    U0s = O.init_params(ushapes, seed=51, random_affine=True)
    model, mblocks = ref_unet(clone_params(U0s))
    ema, eblocks = ref_unet(clone_params(U0s))
    for p in ema.parameters():
        p.detach_()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=0.0001)
    vol, lab = O.synth_acdc_batch(8, shape=(64, 64), seed=78)
    rngd = np.random.default_rng(52)
    traj, boxes, dropbits = [], [], []
    for it in range(2):
        img_a, img_b, uimg_a, uimg_b = vol[:2], vol[2:4], vol[4:6], vol[6:8]
        lab_a, lab_b = lab[:2], lab[2:4]
        d = {k: unet_drop_masks(rngd, 2, (64, 64)) for k in ("t_a", "t_b", "s_unl", "s_l")}
        dropbits.append([np.concatenate([np.packbits(d[k][f"d{i}"].numpy().astype(np.uint8)) for i in range(5)])
                         for k in ("t_a", "t_b", "s_unl", "s_l")])
        with torch.no_grad():
            set_drop_unet(eblocks, d["t_a"])
            pre_a = ema(uimg_a)
            set_drop_unet(eblocks, d["t_b"])
            pre_b = ema(uimg_b)
            plab_a = ref_acdc.get_ACDC_masks(pre_a, nms=1)
            plab_b = ref_acdc.get_ACDC_masks(pre_b, nms=1)
            w, h = int(rngd.integers(0, 64 - 42)), int(rngd.integers(0, 64 - 42))
            box = (w, h, 42, 42)
            boxes.append(box)
            img_mask, loss_mask = O.box_to_mask(box, (64, 64), 2)
        net_input_unl = uimg_a * img_mask + img_a * (1 - img_mask)
        net_input_l = img_b * img_mask + uimg_b * (1 - img_mask)
        set_drop_unet(mblocks, d["s_unl"])
        out_unl = model(net_input_unl)
        set_drop_unet(mblocks, d["s_l"])
        out_l = model(net_input_l)
        unl_dice, unl_ce = ref_acdc.mix_loss(out_unl, plab_a, lab_a, loss_mask, u_weight=0.5, unlab=True)
        l_dice, l_ce = ref_acdc.mix_loss(out_l, lab_b, plab_b, loss_mask, u_weight=0.5)
        loss_ce = unl_ce + l_ce
        loss_dice = unl_dice + l_dice
        loss = (loss_dice + loss_ce) / 2
        opt.zero_grad()
        loss.backward()
        opt.step()
        ref_acdc.update_model_ema(model, ema, 0.99)
        traj.append([loss.item(), loss_dice.item(), loss_ce.item(), float(plab_a.sum()), float(plab_b.sum())])
    sdm, sde = model.state_dict(), ema.state_dict()
    np.savez_compressed(os.path.join(OUT, "acdc_traj.npz"), traj=np.array(traj), boxes=np.array(boxes),
                        dropbits=np.array(dropbits),
                        final_w_stats=np.array([stats(sdm[k]) for k in meta["unet_param_names"][:40]]),
                        final_ema_stats=np.array([stats(sde[k]) for k in meta["unet_param_names"][:40]]),
                        final_ema_rm=sde["encoder.in_conv.conv_conv.1.running_mean"].numpy())
    meta["acdc_traj"] = {"param_seed": 51, "data_seed": 78, "shape": [64, 64], "steps": 2}

    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("golden fixtures written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print(f"  {fn}: {os.path.getsize(os.path.join(OUT, fn))} B")


if __name__ == "__main__":
    main()
