"""ORACLE — test infrastructure only (never imported by the product package).

CPU restatement (plain PyTorch CPU ops, fp32 or fp64) of the two backbones on the FFC hot path,
driven by a flat state dict that uses the reference's parameter names:
  * iResNet      — reference model/resnet_arcface.py:26-55 (IBasicBlock), :58-152 (IResNet)
  * MobileFaceNet — reference model/mobilefacenet_def.py:18-25, :27-52, :55-74, :77-123
  * ResNet (torchvision style, Bottleneck) — reference model/resnet_std.py:55-104, :106-206
Both run in training mode (batch statistics, running-stat update with momentum 0.1, eps 1e-5),
because the reference never calls .eval() on either net (ffc.py:22-23, main.py:116-121).

Pinned against the reference itself through the whole-step fixtures tests/golden/step_*.npz, produced by
importing the reference modules in the build container (tests/golden/make_golden.py) and replayed by
tests/test_oracle_golden.py.

`emulate_bf16=True` keeps the same float64 arithmetic but rounds to bfloat16 at exactly the points where the
MI355X path stores a tensor in bf16 (csrc/iresnet.cpp, csrc/mobilenet.cpp): the image, MFMA-convolution weights,
every convolution output (BatchNorm statistics are taken from the rounded tensor, as the conv epilogue does), every
BatchNorm/PReLU/residual output, and in the backward pass every activation gradient (convolution input gradients,
BatchNorm input gradients, the gradient entering the fc / linear1 contraction).  What is left between the two is
summation order and fp32-vs-fp64 accumulation, so whole-step GRADIENTS can be compared tightly
(tests/test_step_gpu.py) instead of through the bf16 noise band.  With the flag off nothing changes.
"""
import math

import torch
import torch.nn.functional as F

IRESNET_LAYERS = {"ir18": (2, 2, 2, 2), "ir34": (3, 4, 6, 3), "ir50": (3, 4, 14, 3), "ir100": (3, 13, 30, 3),
                  "ir200": (6, 26, 60, 6)}
MOBILE_SETTING = ((2, 64, 5, 2), (4, 128, 1, 2), (2, 128, 6, 1), (4, 128, 1, 2), (2, 128, 2, 1))


# ----------------------------------------------------------------------------------------------
# state-dict construction (names/shapes/initial distributions of the reference constructors)
# ----------------------------------------------------------------------------------------------
def _bn(sd, name, c):
    sd[name + ".weight"] = torch.ones(c)
    sd[name + ".bias"] = torch.zeros(c)
    sd[name + ".running_mean"] = torch.zeros(c)
    sd[name + ".running_var"] = torch.ones(c)
    sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def _uniform(shape, bound, gen):
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def iresnet_state(layers, feat_dim=512, gen=None):
    """resnet_arcface.py:74-105: conv ~ N(0, 0.1), BN 1/0, PReLU 0.25, Linear default init."""
    sd = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = torch.randn(co, ci, k, k, generator=gen) * 0.1

    conv("conv1", 64, 3, 3)
    _bn(sd, "bn1", 64)
    sd["prelu.weight"] = torch.full((64,), 0.25)
    cin = 64
    for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), layers), start=1):
        for bi in range(nblk):
            pre = "layer%d.%d" % (li, bi)
            _bn(sd, pre + ".bn1", cin)
            conv(pre + ".conv1", planes, cin, 3)
            _bn(sd, pre + ".bn2", planes)
            sd[pre + ".prelu.weight"] = torch.full((planes,), 0.25)
            conv(pre + ".conv2", planes, planes, 3)
            _bn(sd, pre + ".bn3", planes)
            if bi == 0:
                conv(pre + ".downsample.0", planes, cin, 1)
                _bn(sd, pre + ".downsample.1", planes)
            cin = planes
    _bn(sd, "bn2", 512)
    fan_in = 512 * 49
    sd["fc.weight"] = _uniform((feat_dim, fan_in), 1.0 / math.sqrt(fan_in), gen)
    sd["fc.bias"] = _uniform((feat_dim,), 1.0 / math.sqrt(fan_in), gen)
    _bn(sd, "features", feat_dim)
    return sd


def mobilefacenet_state(feat_dim=128, gen=None):
    """mobilefacenet_def.py:77-102: nn.Conv2d default init (kaiming-uniform a=sqrt 5 ⇒ U(±1/sqrt(fan_in)))."""
    sd = {}

    def convblock(name, ci, co, k, dw=False, linear=False):
        fan_in = (1 if dw else ci) * k * k
        sd[name + ".conv.weight"] = _uniform((co, 1 if dw else ci, k, k), 1.0 / math.sqrt(fan_in), gen)
        _bn(sd, name + ".bn", co)
        if not linear:
            sd[name + ".prelu.weight"] = torch.full((co,), 0.25)

    convblock("conv1", 3, 64, 3)
    convblock("dw_conv1", 64, 64, 3, dw=True)
    cur, bi = 64, 0
    for t, c, n, s in MOBILE_SETTING:
        for i in range(n):
            pre = "blocks.%d.conv" % bi
            mid = cur * t
            sd[pre + ".0.weight"] = _uniform((mid, cur, 1, 1), 1.0 / math.sqrt(cur), gen)
            _bn(sd, pre + ".1", mid)
            sd[pre + ".2.weight"] = torch.full((mid,), 0.25)
            sd[pre + ".3.weight"] = _uniform((mid, 1, 3, 3), 1.0 / 3.0, gen)
            _bn(sd, pre + ".4", mid)
            sd[pre + ".5.weight"] = torch.full((mid,), 0.25)
            sd[pre + ".6.weight"] = _uniform((c, mid, 1, 1), 1.0 / math.sqrt(mid), gen)
            _bn(sd, pre + ".7", c)
            cur = c
            bi += 1
    convblock("conv2", 128, 512, 1)
    convblock("linear7", 512, 512, 7, dw=True, linear=True)
    convblock("linear1", 512, feat_dim, 1, linear=True)
    return sd


RESNET_LAYERS = {"r50": (3, 4, 6, 3), "r101": (3, 4, 23, 3), "rtiny": (1, 1, 1, 1)}


def resnet_std_state(layers, feat_dim=512, gen=None, image_size=224):
    """resnet_std.py:127-151: kaiming_normal(fan_out, relu) convolutions, BN 1/0, Linear default init."""
    sd = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = torch.randn(co, ci, k, k, generator=gen) * math.sqrt(2.0 / (co * k * k))

    conv("conv1", 64, 3, 7)
    _bn(sd, "bn1", 64)
    cin = 64
    for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), layers), start=1):
        for bi in range(nblk):
            pre = "layer%d.%d" % (li, bi)
            conv(pre + ".conv1", planes, cin, 1)
            _bn(sd, pre + ".bn1", planes)
            conv(pre + ".conv2", planes, planes, 3)
            _bn(sd, pre + ".bn2", planes)
            conv(pre + ".conv3", planes * 4, planes, 1)
            _bn(sd, pre + ".bn3", planes * 4)
            if bi == 0:
                conv(pre + ".downsample.0", planes * 4, cin, 1)
                _bn(sd, pre + ".downsample.1", planes * 4)
            cin = planes * 4
    side = image_size // 32
    fan_in = cin * side * side
    sd["fc.weight"] = _uniform((feat_dim, fan_in), 1.0 / math.sqrt(fan_in), gen)
    sd["fc.bias"] = _uniform((feat_dim,), 1.0 / math.sqrt(fan_in), gen)
    _bn(sd, "features", feat_dim)
    return sd


def is_buffer(name):
    return name.endswith("running_mean") or name.endswith("running_var") or name.endswith("num_batches_tracked")


def trainable(name, net_type=None):
    """resnet_arcface.py:97-98 freezes features.weight of the iResNet (resnet_std.py does not: pass its net_type);
    buffers never train."""
    if net_type in RESNET_LAYERS:
        return not is_buffer(name)
    return not is_buffer(name) and name != "features.weight"


# ----------------------------------------------------------------------------------------------
# bf16 storage emulation (see module docstring)
# ----------------------------------------------------------------------------------------------
def _bf16(t):
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _RoundBoth(torch.autograd.Function):
    """A tensor the device stores in bf16: rounded in the forward pass, its gradient rounded in the backward pass."""

    @staticmethod
    def forward(ctx, x):
        return _bf16(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


class _RoundGrad(torch.autograd.Function):
    """Identity whose gradient is rounded: an input gradient the device writes to a bf16 buffer before it is summed
    with another contribution."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


class Emu(object):
    def __init__(self, on):
        self.on = bool(on)

    def R(self, x):
        return _RoundBoth.apply(x) if self.on else x

    def G(self, x):
        return _RoundGrad.apply(x) if self.on else x

    def W(self, w):          # bf16 operand copy of an fp32 master weight (gradient goes to the master)
        return w + (_bf16(w.detach()) - w.detach()) if self.on else w

    def X(self, x):          # the input image (no gradient)
        return _bf16(x) if self.on else x


# ----------------------------------------------------------------------------------------------
# forward passes (training-mode BN; running stats updated in place in `sd`)
# ----------------------------------------------------------------------------------------------
def _bn_train(sd, name, x):
    rm, rv = sd[name + ".running_mean"], sd[name + ".running_var"]
    y = F.batch_norm(x, rm, rv, sd[name + ".weight"], sd[name + ".bias"], True, 0.1, 1e-5)
    sd[name + ".num_batches_tracked"] += 1
    return y


def iresnet_block(sd, h, pre, first, E):
    """IBasicBlock.forward (resnet_arcface.py:44-55): BN -> 3x3 -> BN -> PReLU -> 3x3(stride) -> BN, plus the identity or
    (first block of a stage: stride 2) the 1x1-s2 + BN shortcut (:112-136)."""
    conv = lambda t, name, stride, pad: E.R(F.conv2d(E.G(t), E.W(sd[name + ".weight"]), None, stride, pad))
    stride = 2 if first else 1
    o = E.R(_bn_train(sd, pre + ".bn1", h))
    o = conv(o, pre + ".conv1", 1, 1)
    o = E.R(F.prelu(_bn_train(sd, pre + ".bn2", o), sd[pre + ".prelu.weight"]))
    o = conv(o, pre + ".conv2", stride, 1)
    o = _bn_train(sd, pre + ".bn3", o)
    if first:
        idn = conv(h, pre + ".downsample.0", stride, 0)
        idn = E.R(_bn_train(sd, pre + ".downsample.1", idn))
    else:
        idn = h
    return E.R(o + idn)


def iresnet_forward(sd, x, layers, emulate_bf16=False):
    E = Emu(emulate_bf16)
    conv = lambda h, name, stride, pad: E.R(F.conv2d(E.G(h), E.W(sd[name + ".weight"]), None, stride, pad))
    h = conv(E.X(x), "conv1", 1, 1)
    h = E.R(F.prelu(_bn_train(sd, "bn1", h), sd["prelu.weight"]))
    for li, nblk in enumerate(layers, start=1):
        for bi in range(nblk):
            h = iresnet_block(sd, h, "layer%d.%d" % (li, bi), bi == 0, E)
    h = E.R(_bn_train(sd, "bn2", h))
    h = torch.flatten(h, 1)
    h = E.G(F.linear(h, E.W(sd["fc.weight"]), sd["fc.bias"]))
    h = _bn_train(sd, "features", h)
    return F.normalize(h)


def resnet_std_forward(sd, x, layers, emulate_bf16=False):
    """resnet_std.py:184-203 (_forward_impl) with Bottleneck.forward (:82-104)."""
    E = Emu(emulate_bf16)
    conv = lambda h, name, stride, pad: E.R(F.conv2d(E.G(h), E.W(sd[name + ".weight"]), None, stride, pad))
    h = conv(E.X(x), "conv1", 2, 3)
    h = E.R(F.relu(_bn_train(sd, "bn1", h)))
    h = E.R(F.max_pool2d(h, 3, 2, 1))
    for li, nblk in enumerate(layers, start=1):
        for bi in range(nblk):
            pre = "layer%d.%d" % (li, bi)
            stride = 2 if (bi == 0 and li > 1) else 1
            o = conv(h, pre + ".conv1", 1, 0)
            o = E.R(F.relu(_bn_train(sd, pre + ".bn1", o)))
            o = conv(o, pre + ".conv2", stride, 1)
            o = E.R(F.relu(_bn_train(sd, pre + ".bn2", o)))
            o = conv(o, pre + ".conv3", 1, 0)
            o = _bn_train(sd, pre + ".bn3", o)
            if bi == 0:
                idn = conv(h, pre + ".downsample.0", stride, 0)
                idn = E.R(_bn_train(sd, pre + ".downsample.1", idn))
            else:
                idn = h
            h = E.R(F.relu(o + idn))
    h = torch.flatten(h, 1)
    h = E.G(F.linear(h, E.W(sd["fc.weight"]), sd["fc.bias"]))
    h = _bn_train(sd, "features", h)
    return F.normalize(h)


def mobile_unit(sd, E, h, wname, bn, prelu, stride, pad, dw=False, res=None):
    """conv -> BN (-> PReLU) (+ residual), one executor unit (ConvBlock, mobilefacenet_def.py:55-74): the convolution
    output and the unit output are bf16 tensors on the device; depthwise filters are used as fp32 (csrc/dw.hip), MFMA
    weights as bf16."""
    w = sd[wname]
    c = E.R(F.conv2d(E.G(h), w if dw else E.W(w), None, stride, pad, 1, w.shape[0] if dw else 1))
    a = _bn_train(sd, bn, c)
    if prelu is not None:
        a = F.prelu(a, sd[prelu])
    if res is not None:
        a = a + res
    return E.R(a)


def mobile_bottleneck(sd, h, pre, stride, connect, E):
    """BottleNeck.forward (mobilefacenet_def.py:27-52): 1x1 -> BN -> PReLU -> depthwise 3x3(stride) -> BN -> PReLU -> 1x1 -> BN,
    `x + conv(x)` when stride 1 and equal widths."""
    o = mobile_unit(sd, E, h, pre + ".0.weight", pre + ".1", pre + ".2.weight", 1, 0)
    o = mobile_unit(sd, E, o, pre + ".3.weight", pre + ".4", pre + ".5.weight", stride, 1, dw=True)
    return mobile_unit(sd, E, o, pre + ".6.weight", pre + ".7", None, 1, 0, res=h if connect else None)


def mobilefacenet_forward(sd, x, emulate_bf16=False):
    E = Emu(emulate_bf16)
    unit = lambda *a, **k: mobile_unit(sd, E, *a, **k)
    h = unit(E.X(x), "conv1.conv.weight", "conv1.bn", "conv1.prelu.weight", 2, 1)
    h = unit(h, "dw_conv1.conv.weight", "dw_conv1.bn", "dw_conv1.prelu.weight", 1, 1, dw=True)
    cur, bi = 64, 0
    for t, c, n, s in MOBILE_SETTING:
        for i in range(n):
            stride = s if i == 0 else 1
            h = mobile_bottleneck(sd, h, "blocks.%d.conv" % bi, stride, stride == 1 and cur == c, E)
            cur = c
            bi += 1
    h = unit(h, "conv2.conv.weight", "conv2.bn", "conv2.prelu.weight", 1, 0)
    h = unit(h, "linear7.conv.weight", "linear7.bn", None, 1, 0, dw=True)
    # linear1: the 1x1 contraction runs on MFMA with an fp32 output; its BatchNorm and the normalisation stay fp32
    h = E.G(F.conv2d(E.G(h), E.W(sd["linear1.conv.weight"])))
    h = _bn_train(sd, "linear1.bn", h)
    return F.normalize(torch.flatten(h, 1))


def make_backbone(net_type, feat_dim, gen=None, layers=None, emulate_bf16=False):
    """Returns (state_dict, forward(sd, x))."""
    if net_type == "mobile":
        return mobilefacenet_state(feat_dim, gen), (lambda sd, x: mobilefacenet_forward(sd, x, emulate_bf16))
    if net_type in RESNET_LAYERS:
        lay = tuple(layers) if layers is not None else RESNET_LAYERS[net_type]
        size = 64 if net_type == "rtiny" else 224
        return resnet_std_state(lay, feat_dim, gen, size), (lambda sd, x: resnet_std_forward(sd, x, lay, emulate_bf16))
    lay = tuple(layers) if layers is not None else IRESNET_LAYERS[net_type]
    return iresnet_state(lay, feat_dim, gen), (lambda sd, x: iresnet_forward(sd, x, lay, emulate_bf16))
