"""The reference's on-disk datasets read into a device-resident cache: `LAHeart` (dataloaders/dataset.py:90-126)
and `BaseDataSets` (ACDC, :15-50).  File lists, directory layout, `num` truncation, the split -> transform rule and the returned
sample dicts are the reference's; what differs is WHERE the case lives: every h5 file is read once (h5py, on first use), its
image (float32) and label (uint8) stay in HBM, and the transforms handed in are the device ones
(dataset.DeviceRotFlipCrop / DeviceRandomGenerator: one gather kernel per tensor) -- no worker processes, no per-iteration
host -> device copy of ~10 MB volumes.  288 GB of HBM holds either training set many times over (LA: 80 cases x ~35 MB).

h5py is not part of the image this repository was built in: the readers import it on first use and say so if it is missing;
`read_h5` is the one function that touches it (the tests substitute it to pin the list / split / cache logic)."""
import logging
import os

import numpy as np
import torch
from torch.utils.data import Dataset


def read_h5(path):
    """-> (image ndarray, label ndarray) of one case file (datasets 'image', 'label': dataset.py:42-43, :120-121)"""
    try:
        import h5py
    except ImportError as e:          # loud: there is no fallback format
        raise ImportError("reading the LA / ACDC / pancreas case files needs h5py (not installed here); "
                          "the training scripts fall back to synthetic cases only when no file list exists") from e
    with h5py.File(path, "r") as f:
        return f["image"][:], f["label"][:]


def _lines(path):
    with open(path, "r") as f:
        return [ln.strip() for ln in f.readlines() if ln.strip()]


class _DeviceCache:
    """case files -> (float32 image, uint8 label) device tensors, read once"""

    def __init__(self, device):
        self.device = torch.device(device)
        self._hit = {}

    def get(self, path):
        hit = self._hit.get(path)
        if hit is None:
            from . import h5_datasets as me          # late binding: tests replace read_h5
            image, label = me.read_h5(path)
            hit = (torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).to(self.device),
                   torch.from_numpy(np.ascontiguousarray(label).astype(np.uint8)).to(self.device))
            self._hit[path] = hit
        return hit


class LAHeart(Dataset):
    """LA: `<base>/train.list` | `test.list`, cases at `<base>/2018LA_Seg_Training Set/<name>/mri_norm2.h5` (:90-126).
    transform: a device transform on {'image': [W,H,D] float32, 'label': [W,H,D] uint8} (DeviceRotFlipCrop), or None."""

    def __init__(self, base_dir=None, split="train", num=None, transform=None, device="cpu"):
        self._base_dir = base_dir
        self.transform = transform
        self.image_list = _lines(os.path.join(base_dir, "train.list" if split == "train" else "test.list"))
        if num is not None:
            self.image_list = self.image_list[:num]
        self._cache = _DeviceCache(device)
        logging.info("total {} samples".format(len(self.image_list)))

    def __len__(self):
        return len(self.image_list)

    def case_path(self, idx):
        return self._base_dir + "/2018LA_Seg_Training Set/" + self.image_list[idx] + "/mri_norm2.h5"

    def __getitem__(self, idx):
        image, label = self._cache.get(self.case_path(idx))
        sample = {"image": image, "label": label}
        if self.transform:
            sample = self.transform(sample)
        return sample


class BaseDataSets(Dataset):
    """ACDC: split 'train' -> `<base>/train_slices.list`, slices at `<base>/data/slices/<case>.h5`, transformed;
    split 'val' -> `<base>/val.list`, volumes at `<base>/data/<case>.h5`, untransformed; `num` truncates the training list only;
    every sample carries its 'case' (:15-50)."""

    def __init__(self, base_dir=None, split="train", num=None, transform=None, device="cpu"):
        self._base_dir = base_dir
        self.split = split
        self.transform = transform
        self.sample_list = []
        if split == "train":
            self.sample_list = _lines(os.path.join(base_dir, "train_slices.list"))
        elif split == "val":
            self.sample_list = _lines(os.path.join(base_dir, "val.list"))
        if num is not None and split == "train":
            self.sample_list = self.sample_list[:num]
        self._cache = _DeviceCache(device)
        logging.info("total {} samples".format(len(self.sample_list)))

    def __len__(self):
        return len(self.sample_list)

    def case_path(self, idx):
        case = self.sample_list[idx]
        return self._base_dir + ("/data/slices/{}.h5" if self.split == "train" else "/data/{}.h5").format(case)

    def __getitem__(self, idx):
        image, label = self._cache.get(self.case_path(idx))
        sample = {"image": image, "label": label}
        if self.split == "train":
            sample = self.transform(sample)
        sample["case"] = self.sample_list[idx]
        return sample
