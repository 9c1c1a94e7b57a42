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
