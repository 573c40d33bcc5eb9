"""MobileFaceNet — same module tree / state-dict keys / initialisation as the reference
(model/mobilefacenet_def.py:18-25 settings, :27-52 BottleNeck, :55-74 ConvBlock, :77-123 MobileFaceNet),
executed by the native gfx950 executor (csrc/mobilenet.cpp): pointwise convolutions on MFMA, depthwise
ones on the HBM-bound kernels of csrc/dw.hip.  The torch.nn layers are parameter containers only."""
import ctypes

from torch import nn

from .. import _lib
from ._native import NativeBackbone

# (expansion t, output channels c, repeats n, stride of the first repeat s) — mobilefacenet_def.py:18-25
BOTTLENECKS = ((2, 64, 5, 2), (4, 128, 1, 2), (2, 128, 6, 1), (4, 128, 1, 2), (2, 128, 2, 1))
MobileFaceNet_BottleNeck_Setting = [list(r) for r in BOTTLENECKS]


def _unit(cin, cout, k, stride, pad, depthwise=False, act=True):
    """Parameter container with the reference ConvBlock's attribute names: conv, bn[, prelu]."""
    m = nn.Module()
    m.add_module("conv", nn.Conv2d(cin, cout, k, stride, pad, groups=cin if depthwise else 1, bias=False))
    m.add_module("bn", nn.BatchNorm2d(cout))
    if act:
        m.add_module("prelu", nn.PReLU(cout))
    return m


def _bottleneck(cin, cout, stride, t):
    """Parameter container with the reference BottleNeck's layout: `.conv` = Sequential of
    pointwise / BN / PReLU / depthwise / BN / PReLU / pointwise / BN (indices 0..7)."""
    mid = cin * t
    seq = [nn.Conv2d(cin, mid, 1, bias=False), nn.BatchNorm2d(mid), nn.PReLU(mid),
           nn.Conv2d(mid, mid, 3, stride, 1, groups=mid, bias=False), nn.BatchNorm2d(mid), nn.PReLU(mid),
           nn.Conv2d(mid, cout, 1, bias=False), nn.BatchNorm2d(cout)]
    m = nn.Module()
    m.add_module("conv", nn.Sequential(*seq))
    m.connect = stride == 1 and cin == cout
    return m


class MobileFaceNet(NativeBackbone):
    _cprefix = "vlsfr_mobilenet"

    def __init__(self, feat_dim=128, fp16=False, bottleneck_setting=MobileFaceNet_BottleNeck_Setting):
        super(MobileFaceNet, self).__init__()
        if [list(r) for r in bottleneck_setting] != MobileFaceNet_BottleNeck_Setting:
            raise _lib.VlsfrError("MobileFaceNet: only the reference bottleneck table is covered by the native executor")
        self.feat_dim, self.image_size, self.fp16 = int(feat_dim), 112, fp16
        self.conv1 = _unit(3, 64, 3, 2, 1)
        self.dw_conv1 = _unit(64, 64, 3, 1, 1, depthwise=True)
        width, blocks = 64, []
        for t, c, n, s in BOTTLENECKS:
            for i in range(n):
                blocks.append(_bottleneck(width, c, s if i == 0 else 1, t))
                width = c
        self.cur_channel = width
        self.blocks = nn.Sequential(*blocks)
        self.conv2 = _unit(128, 512, 1, 1, 0)
        self.linear7 = _unit(512, 512, 7, 1, 0, depthwise=True, act=False)
        self.linear1 = _unit(512, feat_dim, 1, 1, 0, act=False)
        # the stem weight is read as [64][3][3][3] = channels_last memory; 1x1 and depthwise weights are
        # the same bytes in either format
        self.conv1.conv.weight.data = self.conv1.conv.weight.data.contiguous(memory_format=__import__("torch").channels_last)
        self._init_native()

    def _tables(self):
        import torch
        params, running = super(MobileFaceNet, self)._tables()
        w = self.conv1.conv.weight          # NativeBackbone keeps every 4-D weight channels_last; depthwise /
        for p in params:                     # 1x1 weights have one layout only, so that is already true for them
            if p.dim() == 4 and p.shape[1] == 1 and not p.data.is_contiguous():
                p.data = p.data.contiguous()
        return params, running

    def _create(self, L, B, h):
        L.vlsfr_mobilenet_create.restype = ctypes.c_int
        _lib.check(L.vlsfr_mobilenet_create(ctypes.c_int32(self.feat_dim), ctypes.c_int32(B), ctypes.c_int32(112),
                                            ctypes.byref(h)), "vlsfr_mobilenet_create")
