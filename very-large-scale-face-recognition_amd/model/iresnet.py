"""iResNet backbone — same module tree / state-dict keys / initialisation as the reference
(model/resnet_arcface.py:26-55 IBasicBlock, :58-152 IResNet, :162-184 factories), executed by the
native gfx950 executor (csrc/iresnet.cpp) instead of per-layer PyTorch ops.

The torch.nn layers below are parameter containers only — they are never called.  Convolution
weights live in channels_last memory ([Cout][R][S][Cin], what the MFMA kernels read); their logical
shape and state-dict layout stay OIHW, so checkpoints interchange with the reference.
"""
import ctypes

import torch
from torch import nn

from .. import _lib


class IBasicBlock(nn.Module):           # resnet_arcface.py:26-55
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(IBasicBlock, self).__init__()
        self.bn1 = nn.BatchNorm2d(inplanes, eps=1e-05)
        self.conv1 = nn.Conv2d(inplanes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, eps=1e-05)
        self.prelu = nn.PReLU(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes, eps=1e-05)
        self.downsample = downsample
        self.stride = stride


from ._native import NativeBackbone


class IResNet(NativeBackbone):
    fc_scale = 7 * 7
    _cprefix = "vlsfr_iresnet"

    def __init__(self, layers, dropout=0, feat_dim=512, fp16=False, image_size=112):
        super(IResNet, self).__init__()
        if dropout != 0:
            raise _lib.VlsfrError("IResNet: only dropout=0 (the reference default, resnet_arcface.py:61) is covered")
        self.fp16 = fp16                      # kept for API parity; the executor always computes in bf16/fp32
        self.layers_cfg = tuple(int(v) for v in layers)
        self.feat_dim, self.image_size = int(feat_dim), int(image_size)
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64, eps=1e-05)
        self.prelu = nn.PReLU(64)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1])
        self.layer3 = self._make_layer(256, layers[2])
        self.layer4 = self._make_layer(512, layers[3])
        self.bn2 = nn.BatchNorm2d(512, eps=1e-05)
        self.dropout = nn.Dropout(p=dropout, inplace=True)
        side = image_size // 16
        self.fc = nn.Linear(512 * side * side, feat_dim)
        self.features = nn.BatchNorm1d(feat_dim, eps=1e-05)
        nn.init.constant_(self.features.weight, 1.0)          # resnet_arcface.py:97-98
        self.features.weight.requires_grad = False
        for m in self.modules():                               # resnet_arcface.py:100-105
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.1)
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._init_native()

    def _make_layer(self, planes, blocks):                     # resnet_arcface.py:112-136 (stride 2, no dilation)
        downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, 2, bias=False),
                                   nn.BatchNorm2d(planes, eps=1e-05))
        layers = [IBasicBlock(self.inplanes, planes, 2, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(IBasicBlock(planes, planes))
        return nn.Sequential(*layers)

    # gradient buckets in backward order (csrc/iresnet.cpp vlsfr_iresnet_backward_staged)
    N_BUCKETS = 5
    _BUCKET = {"bn2": 0, "fc": 0, "features": 0, "layer4": 1, "layer3": 2, "layer2": 3, "layer1": 4, "conv1": 4,
               "bn1": 4, "prelu": 4}

    def bucket_of(self, name):
        return self._BUCKET[name.split(".")[0]]

    def _create(self, L, B, h):
        lay = (ctypes.c_int32 * 4)(*self.layers_cfg)
        L.vlsfr_iresnet_create.restype = ctypes.c_int
        _lib.check(L.vlsfr_iresnet_create(lay, ctypes.c_int32(self.feat_dim), ctypes.c_int32(B),
                                          ctypes.c_int32(self.image_size), ctypes.byref(h)), "vlsfr_iresnet_create")


def _iresnet(layers, **kwargs):
    kwargs.pop("pretrained", None)
    kwargs.pop("progress", None)
    return IResNet(layers, **kwargs)


def iresnet18(**kw):
    return _iresnet([2, 2, 2, 2], **kw)


def iresnet34(**kw):
    return _iresnet([3, 4, 6, 3], **kw)


def iresnet50(**kw):
    return _iresnet([3, 4, 14, 3], **kw)


def iresnet100(**kw):
    return _iresnet([3, 13, 30, 3], **kw)


def iresnet200(**kw):
    return _iresnet([6, 26, 60, 6], **kw)
