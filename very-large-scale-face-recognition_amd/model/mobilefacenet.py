"""MobileFaceNet — same module tree / state-dict keys / initialisation as the reference
(model/mobilefacenet_def.py:18-25 settings, :27-52 BottleNeck, :55-74 ConvBlock, :77-123 MobileFaceNet),
executed by the native gfx950 executor (csrc/mobilenet.cpp): pointwise convolutions on MFMA, depthwise
ones on the HBM-bound kernels of csrc/dw.hip.  The torch.nn layers are parameter containers only."""
import ctypes

from torch import nn

from .. import _lib
from ._native import NativeBackbone

MobileFaceNet_BottleNeck_Setting = [
    # t, c , n ,s
    [2, 64, 5, 2],
    [4, 128, 1, 2],
    [2, 128, 6, 1],
    [4, 128, 1, 2],
    [2, 128, 2, 1],
]


class BottleNeck(nn.Module):
    def __init__(self, inp, oup, stride, expansion):
        super(BottleNeck, self).__init__()
        self.connect = stride == 1 and inp == oup
        self.conv = nn.Sequential(
            nn.Conv2d(inp, inp * expansion, 1, 1, 0, bias=False),
            nn.BatchNorm2d(inp * expansion),
            nn.PReLU(inp * expansion),
            nn.Conv2d(inp * expansion, inp * expansion, 3, stride, 1, groups=inp * expansion, bias=False),
            nn.BatchNorm2d(inp * expansion),
            nn.PReLU(inp * expansion),
            nn.Conv2d(inp * expansion, oup, 1, 1, 0, bias=False),
            nn.BatchNorm2d(oup),
        )


class ConvBlock(nn.Module):
    def __init__(self, inp, oup, k, s, p, dw=False, linear=False):
        super(ConvBlock, self).__init__()
        self.linear = linear
        self.conv = nn.Conv2d(inp, oup, k, s, p, groups=inp if dw else 1, bias=False)
        self.bn = nn.BatchNorm2d(oup)
        if not linear:
            self.prelu = nn.PReLU(oup)


class MobileFaceNet(NativeBackbone):
    _cprefix = "vlsfr_mobilenet"

    def __init__(self, feat_dim=128, fp16=False, bottleneck_setting=MobileFaceNet_BottleNeck_Setting):
        super(MobileFaceNet, self).__init__()
        if [list(r) for r in bottleneck_setting] != MobileFaceNet_BottleNeck_Setting:
            raise _lib.VlsfrError("MobileFaceNet: only the reference bottleneck table is covered by the native executor")
        self.feat_dim, self.image_size, self.fp16 = int(feat_dim), 112, fp16
        self.conv1 = ConvBlock(3, 64, 3, 2, 1)
        self.dw_conv1 = ConvBlock(64, 64, 3, 1, 1, dw=True)
        self.cur_channel = 64
        layers = []
        for t, c, n, s in bottleneck_setting:
            for i in range(n):
                layers.append(BottleNeck(self.cur_channel, c, s if i == 0 else 1, t))
                self.cur_channel = c
        self.blocks = nn.Sequential(*layers)
        self.conv2 = ConvBlock(128, 512, 1, 1, 0)
        self.linear7 = ConvBlock(512, 512, 7, 1, 0, dw=True, linear=True)
        self.linear1 = ConvBlock(512, feat_dim, 1, 1, 0, linear=True)
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
