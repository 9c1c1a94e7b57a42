"""Sampler semantics of the reference's dataloaders/dataset.py plus synthetic stand-in datasets
(no h5 files ship with the build).  `TwoStreamBatchSampler` reproduces :280-307 / :340-355: every batch =
`primary_batch_size` labeled indices (one pass over a permutation per epoch) followed by
`secondary_batch_size` unlabeled indices (endless permutations); `len` = n_labeled // primary_batch_size.
The draws come from the global np.random stream, interleaved with the box draws, as in the reference."""
import itertools

import numpy as np
import torch
from torch.utils.data import Dataset
from torch.utils.data.sampler import Sampler

from .. import synth


def iterate_once(iterable):
    return np.random.permutation(iterable)


def iterate_eternally(indices):
    def infinite_shuffles():
        while True:
            yield np.random.permutation(indices)
    return itertools.chain.from_iterable(infinite_shuffles())


def grouper(iterable, n):
    args = [iter(iterable)] * n
    return zip(*args)


class TwoStreamBatchSampler(Sampler):
    def __init__(self, primary_indices, secondary_indices, batch_size, secondary_batch_size):
        self.primary_indices = primary_indices
        self.secondary_indices = secondary_indices
        self.secondary_batch_size = secondary_batch_size
        self.primary_batch_size = batch_size - secondary_batch_size
        assert len(self.primary_indices) >= self.primary_batch_size > 0
        assert len(self.secondary_indices) >= self.secondary_batch_size > 0

    def __iter__(self):
        primary_iter = iterate_once(self.primary_indices)
        secondary_iter = iterate_eternally(self.secondary_indices)
        return (primary_batch + secondary_batch for (primary_batch, secondary_batch)
                in zip(grouper(primary_iter, self.primary_batch_size), grouper(secondary_iter, self.secondary_batch_size)))

    def __len__(self):
        return len(self.primary_indices) // self.primary_batch_size


class SyntheticLA(Dataset):
    """80 synthetic LA-like cases (112x112x80 crops), generated once and kept on the device"""

    def __init__(self, num=80, shape=(112, 112, 80), device="cpu", seed=1337, distinct=8):
        vols, labs = synth.la_batch(distinct, shape=shape, seed=seed)
        self.vols, self.labs, self.num, self.distinct = vols.to(device), labs.to(device), num, distinct

    def __len__(self):
        return self.num

    def __getitem__(self, idx):
        j = idx % self.distinct
        return {"image": self.vols[j], "label": self.labs[j]}


class SyntheticACDC(Dataset):
    def __init__(self, num=1312, shape=(256, 256), device="cpu", seed=1337, distinct=32):
        vols, labs = synth.acdc_batch(distinct, shape=shape, seed=seed)
        self.vols, self.labs, self.num, self.distinct = vols.to(device), labs.to(device), num, distinct

    def __len__(self):
        return self.num

    def __getitem__(self, idx):
        j = idx % self.distinct
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
