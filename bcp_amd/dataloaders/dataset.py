"""Sampler semantics of the reference's dataloaders/dataset.py plus synthetic stand-in datasets
(no h5 files ship with the build).  `TwoStreamBatchSampler` reproduces :280-307 / :340-355: every batch =
`primary_batch_size` labeled indices (one pass over a permutation per epoch) followed by
`secondary_batch_size` unlabeled indices (endless permutations); `len` = n_labeled // primary_batch_size.
The draws come from the global np.random stream, interleaved with the box draws, as in the reference."""
import numpy as np
import torch
from torch.utils.data import Dataset
from torch.utils.data.sampler import Sampler

from .. import synth


class _EndlessPermutations:
    """the unlabeled stream: indices served from one np.random.permutation after another; a new permutation is drawn only when
    the current one runs out (so a batch may straddle two of them), exactly when the reference's lazy generator chain draws it"""

    def __init__(self, indices):
        self.indices, self.pending = indices, []

    def take(self, n):
        out = []
        while len(out) < n:
            if not self.pending:
                self.pending = list(np.random.permutation(self.indices))
            room = n - len(out)
            out.extend(self.pending[:room])
            self.pending = self.pending[room:]
        return tuple(out)


class TwoStreamBatchSampler(Sampler):
    """dataloaders/dataset.py:280-307 (+ helpers :340-355): batch = `batch_size - secondary_batch_size` labeled indices then
    `secondary_batch_size` unlabeled ones; an epoch = one pass over ONE permutation of the labeled indices (a remainder that does
    not fill a batch is dropped), the unlabeled permutations restart with every epoch.  RNG order per epoch (global np.random):
    the labeled permutation when iteration starts, the first unlabeled permutation when the first batch is assembled.
    Pinned against the reference's class on seeded streams (tests/golden/sampler.npz)."""

    def __init__(self, primary_indices, secondary_indices, batch_size, secondary_batch_size):
        self.primary_indices, self.secondary_indices = primary_indices, secondary_indices
        self.secondary_batch_size = secondary_batch_size
        self.primary_batch_size = batch_size - secondary_batch_size
        if not 0 < self.primary_batch_size <= len(primary_indices):
            raise AssertionError("not enough labeled samples for one batch")
        if not 0 < self.secondary_batch_size <= len(secondary_indices):
            raise AssertionError("not enough unlabeled samples for one batch")

    def __len__(self):
        return len(self.primary_indices) // self.primary_batch_size

    def __iter__(self):
        labeled = np.random.permutation(self.primary_indices)
        unlabeled = _EndlessPermutations(self.secondary_indices)
        pb = self.primary_batch_size
        for k in range(len(self)):
            yield tuple(labeled[k * pb:(k + 1) * pb]) + unlabeled.take(self.secondary_batch_size)


class SyntheticLA(Dataset):
    """80 synthetic LA-like cases, generated once and kept on the device.  Without a transform the cases already have the
    training patch shape (112x112x80); with one (DeviceRotFlipCrop: the reference's RandomRotFlip -> RandomCrop -> ToTensor,
    LA_BCP_train.py:122-126) they are generated at `raw_shape` and every __getitem__ runs the transform on the device."""

    def __init__(self, num=80, shape=(112, 112, 80), device="cpu", seed=1337, distinct=8, transform=None, raw_shape=None):
        vols, labs = synth.la_batch(distinct, shape=raw_shape if (transform is not None and raw_shape) else shape, seed=seed)
        self.vols, self.labs, self.num, self.distinct = vols.to(device), labs.to(device), num, distinct
        self.transform = transform
        self.labs8 = self.labs.to(torch.uint8) if transform is not None else None

    def __len__(self):
        return self.num

    def __getitem__(self, idx):
        j = idx % self.distinct
        if self.transform is not None:
            return self.transform({"image": self.vols[j, 0], "label": self.labs8[j]})
        return {"image": self.vols[j], "label": self.labs[j]}


class SyntheticACDC(Dataset):
    """synthetic ACDC-like slices on the device; with a transform (DeviceRandomGenerator: the reference's RandomGenerator,
    ACDC_BCP_train.py:209-211) they are generated at `raw_shape` and zoomed to the training resolution per __getitem__."""

    def __init__(self, num=1312, shape=(256, 256), device="cpu", seed=1337, distinct=32, transform=None, raw_shape=None):
        vols, labs = synth.acdc_batch(distinct, shape=raw_shape if (transform is not None and raw_shape) else shape, seed=seed)
        self.vols, self.labs, self.num, self.distinct = vols.to(device), labs.to(device), num, distinct
        self.transform = transform
        self.labs8 = self.labs.to(torch.uint8) if transform is not None else None

    def __len__(self):
        return self.num

    def __getitem__(self, idx):
        j = idx % self.distinct
        if self.transform is not None:
            return self.transform({"image": self.vols[j, 0], "label": self.labs8[j]})
        return {"image": self.vols[j], "label": self.labs[j]}


def batches(dataset, batch_sampler):
    """DataLoader(batch_sampler=...) without worker processes: the synthetic data already lives on the device"""
    for idxs in batch_sampler:
        items = [dataset[int(i)] for i in idxs]
        yield {"image": torch.stack([it["image"] for it in items]), "label": torch.stack([it["label"] for it in items])}


class DeviceRotFlipCrop:
    """RandomRotFlip() -> RandomCrop(output_size) -> ToTensor() of the reference's LA pipeline (dataloaders/dataset.py:52-59,
    173-214, 263-273; LA_BCP_train.py:122-126) on a device-resident case: the random draws come from np.random in the
    reference's order (k, axis, then w1, h1, d1 on the padded shape), the data movement is ONE gather kernel per tensor
    (csrc/eval.hip k_crop_rotflip) -- no host copy of the volume, no intermediate rotated / flipped / padded copies.

    sample: {'image': float32 [W,H,D] device tensor, 'label': uint8 [W,H,D]} -> {'image': [1,P0,P1,P2] float32, 'label': [P0,P1,P2] uint8}"""

    def __init__(self, output_size):
        self.output_size = tuple(int(v) for v in output_size)

    def draw(self, shape):
        """the five np.random draws of random_rot_flip + RandomCrop.__call__ for a volume of `shape`"""
        k = int(np.random.randint(0, 4))
        axis = int(np.random.randint(0, 2))
        rs = (shape[1], shape[0], shape[2]) if k % 2 else tuple(shape)            # shape after rot90 in the (0, 1) plane
        P = self.output_size
        if rs[0] <= P[0] or rs[1] <= P[1] or rs[2] <= P[2]:                       # dataset.py:190-198
            pads = tuple(max((P[i] - rs[i]) // 2 + 3, 0) for i in range(3))
        else:
            pads = (0, 0, 0)
        w, h, d = (rs[i] + 2 * pads[i] for i in range(3))
        w1 = int(np.random.randint(0, w - P[0]))
        h1 = int(np.random.randint(0, h - P[1]))
        d1 = int(np.random.randint(0, d - P[2]))
        return k, axis, pads, (w1, h1, d1)

    def __call__(self, sample):
        from ..utils.BCP_utils import _cpu_ops
        from ..hip_ops import Ops
        image, label = sample["image"], sample["label"]
        ops = Ops.product() if image.is_cuda else _cpu_ops()
        k, axis, pads, org = self.draw(tuple(image.shape))
        img = ops.crop_rotflip(image.contiguous(), self.output_size, k, axis, pads, org)
        lab = ops.crop_rotflip(label.contiguous(), self.output_size, k, axis, pads, org)
        return {"image": img.unsqueeze(0), "label": lab}


class DeviceRandomGenerator:
    """RandomGenerator(output_size) of the reference's ACDC pipeline (dataloaders/dataset.py:69-88; ACDC_BCP_train.py:209-211) on
    a device-resident slice: with probability 1/2 rot90 + flip (np.random k, axis), else with probability 1/2 a rotation by a
    whole number of degrees in [-20, 20) (scipy.ndimage.rotate, order=0), then the nearest-neighbour zoom to `output_size`.
    The draws come from python's `random` and `np.random` in the reference's order; the data movement is ONE gather kernel per
    tensor (csrc/eval.hip k_acdc_augment) that restates scipy's order-0 coordinate arithmetic in fp64 -- no host copy of the
    slice, no intermediate rotated / zoomed arrays.

    sample: {'image': float32 [H,W] device tensor, 'label': uint8 [H,W]} -> {'image': [1,OH,OW] float32, 'label': [OH,OW] uint8}"""

    def __init__(self, output_size):
        self.output_size = tuple(int(v) for v in output_size)

    @staticmethod
    def draw(shape):
        """-> (mode, k, axis, affine6): the random.random / np.random draws of RandomGenerator.__call__"""
        import random
        if random.random() > 0.5:
            k = int(np.random.randint(0, 4))
            axis = int(np.random.randint(0, 2))
            return 1, k, axis, None
        if random.random() > 0.5:
            from scipy import special          # the two functions scipy.ndimage.rotate builds its matrix from
            angle = int(np.random.randint(-20, 20))
            c, s = float(special.cosdg(angle)), float(special.sindg(angle))
            m = np.array([[c, s], [-s, c]])
            ctr = (np.asarray(shape, dtype=np.float64) - 1) / 2
            off = ctr - m @ ctr
            return 2, 0, 0, [m[0, 0], m[0, 1], m[1, 0], m[1, 1], off[0], off[1]]
        return 0, 0, 0, None

    def __call__(self, sample):
        from ..utils.BCP_utils import _cpu_ops
        from ..hip_ops import Ops
        image, label = sample["image"], sample["label"]
        ops = Ops.product() if image.is_cuda else _cpu_ops()
        mode, k, axis, aff = self.draw(tuple(image.shape))
        img = ops.acdc_augment(image.contiguous(), self.output_size, mode, k, axis, aff)
        lab = ops.acdc_augment(label.contiguous(), self.output_size, mode, k, axis, aff)
        return {"image": img.unsqueeze(0), "label": lab}

