"""Counterparts of the hot-path helpers in the reference's code/pancreas/pancreas_utils.py."""
import numpy as np
import torch

from ..train_step import get_cut_mask  # noqa: F401  (:275-281; connect_mode=2 -> 18-connectivity)
from ..utils import BCP_utils as BU
from ..utils.BCP_utils import update_ema_variables  # noqa: F401  (:299-302)


def generate_mask(img, patch_size):
    """:187-200 -- one patch_size^3 zero box in a 96^3 ones volume, three np.random.randint draws (w, h, z)"""
    origin = tuple(int(np.random.randint(0, 96 - patch_size)) for _ in range(3))      # w, h, z: drawn in this order
    box = origin + (patch_size,) * 3
    vol = (96, 96, 96)
    return BU.BoxMask(box, vol, None, False, img.device), BU.BoxMask(box, vol, img.shape[0], False, img.device)


def save_net_opt(net, optimizer, path, epoch):
    """:160-166 -- {'net', 'opt', 'epoch'}"""
    torch.save({"net": net.state_dict(), "opt": optimizer.state_dict(), "epoch": epoch}, str(path))


def load_net_opt(net, optimizer, path):
    """:169-172"""
    state = torch.load(str(path))
    net.load_state_dict(state["net"])
    optimizer.load_state_dict(state["opt"])


def save_net(net, path):
    """:175-179"""
    torch.save({"net": net.state_dict()}, str(path))


def load_net(net, path):
    """:182-184"""
    net.load_state_dict(torch.load(str(path))["net"])

