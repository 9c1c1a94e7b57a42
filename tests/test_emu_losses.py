"""Loss drop-ins (utils/losses.py, BCP_utils.mix_loss / sup_loss) on the host simulator vs the reference goldens."""
import numpy as np
import torch

from bcp_amd.utils import BCP_utils as BU
from bcp_amd.utils import losses as L
from test_emu_kernels import emu_ops  # noqa: F401


def test_loss_dropins(emu_ops, golden_dir):
    BU.set_test_ops(emu_ops)
    g = np.load(f"{golden_dir}/mixloss_la.npz")
    a, b, mask = (torch.from_numpy(g[k]) for k in ("a", "b", "mask"))
    lo = torch.from_numpy(g["logits"]).requires_grad_(True)
    l1 = BU.mix_loss(lo, a, b, mask, u_weight=0.5)            # dense-mask path, logical NCDHW logits
    l1.backward()
    assert abs(float(l1.detach()) - float(g["l1"])) < 1e-5
    assert float((lo.grad - torch.from_numpy(g["g1"])).abs().max()) < 1e-7
    bm = BU.BoxMask((2, 3, 1, 10, 10, 5), (16, 16, 8), 2)
    assert abs(float(BU.mix_loss(lo, a, b, bm, u_weight=0.5, unlab=True).detach()) - float(g["l2"])) < 1e-5
    assert abs(float(BU.sup_loss(lo, a).detach()) - float(g["l3"])) < 1e-5
    assert abs(float(L.mask_DiceLoss(2)(lo, a).detach()) - float(g["dice"])) < 1e-5
    ce, dice = L.sup_loss_parts(lo, a)
    assert abs(float(ce.detach()) - float(g["ce"])) < 1e-5 and abs(float(dice.detach()) - float(g["dice"])) < 1e-5
    lo.grad = None
    ((ce + dice) / 2).backward()
    assert float((lo.grad - torch.from_numpy(g["g3"])).abs().max()) < 1e-7
    # BoxMask behaves like the reference's masks
    assert bm.count() == int(mask.sum()) and torch.equal(bm.tensor(), mask)
    assert torch.equal((1 - bm).tensor(), 1 - mask)
    x, y = torch.randn(2, 1, 16, 16, 8), torch.randn(2, 1, 16, 16, 8)
    im = BU.BoxMask((2, 3, 1, 10, 10, 5), (16, 16, 8))
    mixed = x * im + y * (1 - im)
    assert torch.equal(mixed, x * im.tensor(dtype=torch.float32) + y * (1 - im.tensor(dtype=torch.float32)))
    la, lb = torch.randint(0, 2, (2, 16, 16, 8)), torch.randint(0, 2, (2, 16, 16, 8))
    assert torch.equal(la * im + lb * (1 - im), la * im.tensor() + lb * (1 - im.tensor()))


def test_context_mask_draws(golden_dir):
    import json
    meta = json.load(open(f"{golden_dir}/meta.json"))["ops"]
    np.random.seed(1337)
    boxes = [list(BU.context_mask(torch.zeros(2, 1, 112, 112, 80), 2 / 3)[0].box) for _ in range(4)]
    assert boxes == meta["la_boxes_seed1337"]
    from bcp_amd import train_step
    np.random.seed(1337)
    assert [list(train_step.generate_mask(torch.zeros(2, 1, 256, 256))[0].box) for _ in range(4)] == meta["acdc_boxes_seed1337"]
    from bcp_amd.pancreas.pancreas_utils import generate_mask
    np.random.seed(2020)
    assert [list(generate_mask(torch.zeros(1, 1, 96, 96, 96), 64)[0].box) for _ in range(4)] == meta["pancreas_boxes_seed2020"]
