"""Backbone registry — reference model/__init__.py:1-10 (`create_net`: ir50 / r50 / mobile), extended with the deeper /
shallower iResNets the reference defines but does not register (resnet_arcface.py:162-184) and two test-size nets."""
from .. import _lib
from .iresnet import IResNet, iresnet18, iresnet34, iresnet50, iresnet100, iresnet200


def _irtiny(**kwargs):
    """4-block iResNet (one IBasicBlock per stage): the test-size member of the family."""
    return IResNet([1, 1, 1, 1], **kwargs)


def _mobile(**kwargs):
    from .mobilefacenet import MobileFaceNet
    return MobileFaceNet(**kwargs)


def _r50(**kwargs):
    from .resnet_std import resnet50
    return resnet50(**kwargs)


def _rtiny(**kwargs):
    """One Bottleneck per stage at 64 x 64: the test-size member of the torchvision-style family."""
    from .resnet_std import ResNet
    return ResNet([1, 1, 1, 1], image_size=64, **kwargs)


net_creator = {'ir50': iresnet50, 'r50': _r50, 'mobile': _mobile, 'rtiny': _rtiny,
               'irtiny': _irtiny, 'ir18': iresnet18, 'ir34': iresnet34, 'ir100': iresnet100, 'ir200': iresnet200}


def create_net(net_type, **kwargs):
    if net_type not in net_creator:
        raise Exception('Unknown architecture')
    return net_creator[net_type](**kwargs)
