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


def test_h5_dataset_readers_lists_splits_cache(emu_ops, tmp_path, monkeypatch):
    """dataloaders/h5_datasets.py + pancreas/dataloaders.Pancreas: the reference's list files, directory layout, `num` truncation,
    split -> transform rule, `reverse` indexing and __len__ multipliers (dataloaders/dataset.py:15-50, :90-126;
    pancreas/dataloaders.py:110-170); each case file is read ONCE into the cache.  h5py itself is absent here: read_h5 is substituted."""
    import numpy as np
    from bcp_amd.dataloaders import h5_datasets as HD
    from bcp_amd.pancreas import dataloaders as PD
    import pytest
    BU.set_test_ops(emu_ops)        # the crops are the device gather kernel (here: on the simulator)
    reads = []

    def fake_read(path):
        reads.append(path)
        seed = sum(map(ord, path)) % 1000
        rng = np.random.default_rng(seed)
        if "slices" in path:
            return rng.random((20, 24), dtype=np.float32).astype(np.float64), rng.integers(0, 4, (20, 24)).astype(np.int64)
        if "mri_norm2" in path:
            return rng.random((14, 12, 10), dtype=np.float32), rng.integers(0, 2, (14, 12, 10)).astype(np.uint8)
        return rng.random((11, 12, 13), dtype=np.float32), rng.integers(0, 2, (11, 12, 13)).astype(np.uint8)

    monkeypatch.setattr(HD, "read_h5", fake_read)
    base = tmp_path
    (base / "train.list").write_text("caseA\ncaseB\ncaseC\n")
    (base / "test.list").write_text("caseT\n")
    (base / "train_slices.list").write_text("p1_s1\np1_s2\np2_s1\n")
    (base / "val.list").write_text("p9\n")
    # LA
    la = HD.LAHeart(str(base), split="train", num=2)
    assert len(la) == 2 and la.case_path(1).endswith("/2018LA_Seg_Training Set/caseB/mri_norm2.h5")
    s0 = la[0]
    assert s0["image"].dtype == torch.float32 and s0["label"].dtype == torch.uint8 and tuple(s0["image"].shape) == (14, 12, 10)
    la[0]; la[1]; la[0]
    assert len(reads) == 2                                     # cached per case
    assert len(HD.LAHeart(str(base), split="test")) == 1
    seen = []
    la_t = HD.LAHeart(str(base), split="train", transform=lambda s: (seen.append(1), {"image": s["image"][None], "label": s["label"]})[1])
    assert tuple(la_t[2]["image"].shape) == (1, 14, 12, 10) and seen
    # ACDC: train transformed + truncated, val untransformed + not truncated, 'case' everywhere
    tr = HD.BaseDataSets(str(base), split="train", num=2, transform=lambda s: {"image": s["image"][None], "label": s["label"]})
    assert len(tr) == 2 and tr.case_path(0).endswith("/data/slices/p1_s1.h5")
    it = tr[1]
    assert it["case"] == "p1_s2" and tuple(it["image"].shape) == (1, 20, 24) and it["image"].dtype == torch.float32 and it["label"].dtype == torch.uint8
    va = HD.BaseDataSets(str(base), split="val", num=0)
    assert len(va) == 1 and va.case_path(0).endswith("/data/p9.h5") and va[0]["case"] == "p9" and va[0]["image"].dim() == 3
    # pancreas
    ld = base / "lists" / "pancreas" / "10percent"
    ld.mkdir(parents=True)
    (ld / "train_lab.txt").write_text("a.h5\nb.h5\nc.h5\n")
    (ld / "train_unlab.txt").write_text("u1.h5\nu2.h5\n")
    (ld / "test.txt").write_text("t.h5\n")
    np.random.seed(3)
    lab = PD.Pancreas(str(base), "pancreas", "train_lab", list_dir=str(base / "lists"), patch=(8, 8, 8))
    rev = PD.Pancreas(str(base), "pancreas", "train_lab", reverse=True, list_dir=str(base / "lists"), patch=(8, 8, 8))
    unl = PD.Pancreas(str(base), "pancreas", "train_unlab", list_dir=str(base / "lists"), patch=(8, 8, 8))
    assert len(lab) == 30 and len(unl) == 2 and len(PD.Pancreas(str(base), "pancreas", "test", list_dir=str(base / "lists"))) == 1
    n0 = len(reads)
    img, lb = lab[4]                                           # 4 % 3 = 1 -> b.h5
    assert reads[-1].endswith("/b.h5") and tuple(img.shape) == (1, 8, 8, 8) and img.dtype == torch.float32 and lb.dtype == torch.uint8
    rev[4]                                                     # 3 - 1 - 1 = 1 -> b.h5 again, but its own cache
    assert reads[-1].endswith("/b.h5")
    rev[0]
    assert reads[-1].endswith("/c.h5")
    a1, a2 = unl[1], unl[1]                                    # CenterCrop: deterministic
    assert torch.equal(a1[0], a2[0]) and len(reads) == n0 + 4
    # without h5py the real reader says what is missing
    monkeypatch.undo()
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="h5py"):
            HD.read_h5(str(base / "nope.h5"))
