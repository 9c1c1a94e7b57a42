"""Drop-in for the reference's networks/net_factory.py (:5-19): same callables, HIP-backed networks."""
from .VNet import VNet


def _dev():
    import torch
    return torch.device("cuda", torch.cuda.current_device())


def net_factory(net_type="unet", in_chns=1, class_num=2, mode="train", tsne=0):
    if net_type == "unet" and mode == "train":
        from .unet import UNet
        net = UNet(in_chns=in_chns, class_num=class_num).to(_dev())
    elif net_type == "VNet" and mode == "train" and tsne == 0:
        net = VNet(n_channels=in_chns, n_classes=class_num, normalization='batchnorm', has_dropout=True).to(_dev())
    elif net_type == "VNet" and mode == "test" and tsne == 0:
        net = VNet(n_channels=in_chns, n_classes=class_num, normalization='batchnorm', has_dropout=False).to(_dev())
    else:
        raise NotImplementedError(f"net_factory({net_type!r}, mode={mode!r}): not on the BCP hot path (SURVEY.md section 2)")
    return net.flatten_()


def BCP_net(in_chns=1, class_num=2, ema=False):
    from .unet import UNet_2d
    net = UNet_2d(in_chns=in_chns, class_num=class_num).to(_dev()).flatten_()
    if ema:
        for param in net.parameters():
            param.detach_()
    return net
