"""pancreas/Vnet.py:VNet counterpart (InstanceNorm3d, `branchs` head, returns [logits])."""
from ..networks.VNet import VNet as _VNet


class VNet(_VNet):
    def __init__(self, n_channels=1, n_classes=2, n_filters=16, normalization='instancenorm', has_dropout=False):
        assert not has_dropout, "the pancreas scripts build VNet() with the default has_dropout=False"
        super().__init__(n_channels=n_channels, n_classes=n_classes, n_filters=n_filters, normalization=normalization, variant="pancreas")


def create_Vnet(ema=False):
    """pancreas/dataloaders.py:12-19.  The reference wraps the net in nn.DataParallel over two GPUs; here one
    process drives one GPU and data parallelism is bcp_amd/dp.py."""
    import torch
    net = VNet().to(torch.device("cuda", torch.cuda.current_device())).flatten_()
    if ema:
        for param in net.parameters():
            param.detach_()
    return net
