"""torchvision-style ResNet — same module tree / state-dict keys / initialisation as the reference
(model/resnet_std.py:55-104 Bottleneck, :106-206 ResNet, :242-262 factories; `--net_type r50` is the reference's
default, main.py:152), executed by the native gfx950 executor (csrc/resnet.cpp): the 7x7 stem as an im2col GEMM, the
1x1 / 3x3 convolutions on the MFMA kernels of csrc/conv.hip, BatchNorm + ReLU through the BatchNorm kernels (ReLU is a
zero-slope PReLU), max-pool and the post-add ReLU in csrc/resnet_ops.hip.  Input is 224 x 224 (fc is 2048 * 7 * 7
wide, resnet_std.py:140).  The torch.nn layers below are parameter containers only."""
import ctypes

import torch
from torch import nn

from .. import _lib
from ._native import NativeBackbone


class Bottleneck(nn.Module):            # resnet_std.py:55-80 (parameters only)
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__()
        width = planes
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class ResNet(NativeBackbone):
    _cprefix = "vlsfr_resnet"

    def __init__(self, layers, feat_dim=512, fp16=True, image_size=224, zero_init_residual=False):
        super(ResNet, self).__init__()
        self.layers_cfg = tuple(int(v) for v in layers)
        self.feat_dim, self.image_size, self.fp16 = int(feat_dim), int(image_size), fp16
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        side = image_size // 32
        self.fc = nn.Linear(512 * Bottleneck.expansion * side * side, feat_dim)
        self.features = nn.BatchNorm1d(feat_dim, eps=1e-05)
        for m in self.modules():                               # resnet_std.py:146-151
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:                                 # :156-161
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
        self._init_native()

    def _make_layer(self, planes, blocks, stride=1):           # resnet_std.py:169-191 (no dilation)
        downsample = None
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * Bottleneck.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * Bottleneck.expansion))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * Bottleneck.expansion
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _create(self, L, B, h):
        lay = (ctypes.c_int32 * 4)(*self.layers_cfg)
        L.vlsfr_resnet_create.restype = ctypes.c_int
        _lib.check(L.vlsfr_resnet_create(lay, ctypes.c_int32(self.feat_dim), ctypes.c_int32(B),
                                         ctypes.c_int32(self.image_size), ctypes.byref(h)), "vlsfr_resnet_create")


def resnet50(**kw):                     # resnet_std.py:242-251
    kw.pop("pretrained", None)
    kw.pop("progress", None)
    return ResNet([3, 4, 6, 3], **kw)


def resnet101(**kw):                    # resnet_std.py:254-262
    kw.pop("pretrained", None)
    kw.pop("progress", None)
    return ResNet([3, 4, 23, 3], **kw)
