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


# The reference builds the pancreas net as nn.DataParallel(VNet()) (pancreas/dataloaders.py:11-14), so every key of its
# checkpoints carries a "module." prefix.  Ours is a bare network: checkpoints are WRITTEN with the prefix (the reference's
# load_net / load_net_opt take them as they are) and READ with or without it.
def _to_ref_keys(sd):
    return {"module." + k: v for k, v in sd.items()}


def _from_ref_keys(sd):
    if sd and all(k.startswith("module.") for k in sd):
        return {k[len("module."):]: v for k, v in sd.items()}
    return sd


def save_net_opt(net, optimizer, path, epoch):
    """:160-166 -- {'net', 'opt', 'epoch'}; 'opt' in torch.optim's state_dict layout (train_step.FlatAdam.state_dict)"""
    torch.save({"net": _to_ref_keys(net.state_dict()), "opt": optimizer.state_dict(), "epoch": epoch}, str(path))


def load_net_opt(net, optimizer, path):
    """:169-172"""
    state = torch.load(str(path), weights_only=False)
    net.load_state_dict(_from_ref_keys(state["net"]))
    optimizer.load_state_dict(state["opt"])


def save_net(net, path):
    """:175-179"""
    torch.save({"net": _to_ref_keys(net.state_dict())}, str(path))


def load_net(net, path):
    """:182-184"""
    net.load_state_dict(_from_ref_keys(torch.load(str(path), weights_only=False)["net"]))

