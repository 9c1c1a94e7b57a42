"""Device-side counterparts of the transforms in the reference's code/pancreas/dataloaders.py (:22-100): RandomCrop / CenterCrop /
ToTensor on a device-resident case.  The random draws come from np.random in the reference's order (w1, h1, d1 on the padded
shape); the data movement is ONE gather kernel per tensor (csrc/eval.hip k_crop_rotflip with k = 0, flip_axis = -1) -- no host
copy of the volume, no padded intermediate.  `Pancreas` reads the reference's case files (h5, through
dataloaders/h5_datasets.read_h5) into a device-resident cache; `SyntheticPancreas` stands in when no file list exists."""
import numpy as np
import torch

from .Vnet import create_Vnet  # noqa: F401  (:12-19)


def _pads(shape, P):
    """:34-40 / :67-73 -- symmetric zero padding when ANY axis is <= the patch ((P - n) // 2 + 1 per side; the LA flavour adds 3)"""
    if shape[0] <= P[0] or shape[1] <= P[1] or shape[2] <= P[2]:
        return tuple(max((P[i] - shape[i]) // 2 + 1, 0) for i in range(3))
    return (0, 0, 0)


def _gather(samples, P, pads, org):
    from ..utils.BCP_utils import _cpu_ops
    from ..hip_ops import Ops
    out = []
    for s in samples:
        ops = Ops.product() if s.is_cuda else _cpu_ops()
        out.append(ops.crop_rotflip(s.contiguous(), P, 0, -1, pads, org))
    return out


class RandomCrop:
    """:22-61 -- samples: list of [W,H,D] device tensors (float32 image, uint8 label) sharing one crop"""

    def __init__(self, output_size, with_sdf=False):
        self.output_size = tuple(int(v) for v in output_size)

    def draw(self, shape):
        P = self.output_size
        pads = _pads(shape, P)
        w, h, d = (shape[i] + 2 * pads[i] for i in range(3))
        w1 = int(np.random.randint(0, w - P[0]))
        h1 = int(np.random.randint(0, h - P[1]))
        d1 = int(np.random.randint(0, d - P[2]))
        return pads, (w1, h1, d1)

    def __call__(self, samples):
        pads, org = self.draw(tuple(samples[0].shape))
        return _gather(samples, self.output_size, pads, org)


class CenterCrop:
    """:64-91"""

    def __init__(self, output_size):
        self.output_size = tuple(int(v) for v in output_size)

    def draw(self, shape):
        P = self.output_size
        pads = _pads(shape, P)
        return pads, tuple(int(round((shape[i] + 2 * pads[i] - P[i]) / 2.)) for i in range(3))

    def __call__(self, samples):
        pads, org = self.draw(tuple(samples[0].shape))
        return _gather(samples, self.output_size, pads, org)


class ToTensor:
    """:94-101 -- image -> [1,W,H,D] float32; the label stays uint8 on the device until the loss casts it (the reference's
    `.long()` in Pancreas.__getitem__ is a 8x wider copy of the same values)"""

    def __call__(self, sample):
        return [sample[0].reshape((1,) + tuple(sample[0].shape)).to(torch.float32)] + list(sample[1:])


class SyntheticPancreas:
    """stand-in for the reference's `Pancreas` h5 dataset (:110-170): synthetic pancreas-like cases resident on the device, the
    split -> transform table of the reference (train_lab: RandomCrop(96^3); train_unlab / test: CenterCrop(96^3)), its `reverse`
    indexing (:160-162) and its `__len__` multipliers (:151-157)."""

    def __init__(self, split, device="cpu", n_cases=4, raw_shape=(104, 100, 98), patch=(96, 96, 96), labelp=10, reverse=False,
                 seed=2020):
        from .. import synth
        vols, labs = synth.la_batch(n_cases, shape=raw_shape, seed=seed + {"train_lab": 0, "train_unlab": 1}.get(split, 2))
        self.vols, self.labs8 = vols[:, 0].to(device), labs.to(torch.uint8).to(device)
        self.split, self.reverse, self.labelp = split, reverse, labelp
        self.crop = RandomCrop(patch) if split == "train_lab" else CenterCrop(patch)
        self.to_tensor = ToTensor()

    def __len__(self):
        if self.split == "train_lab":
            return len(self.vols) * (5 if self.labelp == 20 else 10)
        return len(self.vols)

    def __getitem__(self, idx):
        n = len(self.vols)
        j = (n - idx % n - 1) if self.reverse else idx % n
        image, label = self.to_tensor(self.crop([self.vols[j], self.labs8[j]]))
        return image, label


def get_dataset_path(list_dir, dataset="pancreas", labelp="10percent"):
    """:103-106 -- the three list files of a split; the reference hard-codes its own home directory, here it is an argument"""
    return ["/".join([str(list_dir), dataset, labelp, f]) for f in ("train_lab.txt", "train_unlab.txt", "test.txt")]


class Pancreas:
    """the reference's `Pancreas` dataset (:110-170) over a device-resident cache: list file per split, `<base_dir>/<line>` case
    paths, split -> transform (train_lab: RandomCrop(96^3); train_unlab / test: CenterCrop(96^3)), `reverse` indexing, the
    `__len__` multipliers of the labeled split (x10 at 10 %, x5 at 20 %).  Returns (image [1,96,96,96] float32, label uint8)."""

    def __init__(self, base_dir, name, split, no_crop=False, labelp=10, reverse=False, TTA=False, list_dir=None, device="cpu",
                 patch=(96, 96, 96)):
        from ..dataloaders.h5_datasets import _DeviceCache, _lines
        self._base_dir = str(base_dir)
        self.split, self.reverse = split, reverse
        self.labelp = "20percent" if labelp == 20 else "10percent"
        paths = get_dataset_path(list_dir if list_dir is not None else self._base_dir, name, self.labelp)
        data_path = paths[0] if split == "train_lab" else (paths[1] if split == "train_unlab" else paths[2])
        self.crop = RandomCrop(patch) if split == "train_lab" else CenterCrop(patch)
        self.to_tensor = ToTensor()
        self.image_list = [self._base_dir + "/{}".format(item) for item in _lines(data_path)]
        self._cache = _DeviceCache(device)
        print("Split : {}, total {} samples".format(split, len(self.image_list)))

    def __len__(self):
        if self.split == "train_lab":
            return len(self.image_list) * (5 if self.labelp == "20percent" else 10)
        return len(self.image_list)

    def __getitem__(self, idx):
        n = len(self.image_list)
        path = self.image_list[n - idx % n - 1] if self.reverse else self.image_list[idx % n]
        image, label = self._cache.get(path)
        image_, label_ = self.to_tensor(self.crop([image, label]))
        return image_, label_
