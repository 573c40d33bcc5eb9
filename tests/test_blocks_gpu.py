"""Executor wiring pinned tighter than the whole-network bf16 band (tests/test_step_gpu.py): ONE block of a backbone,
teacher-forced.  The float64 oracle block (oracle/backbones_ref.py iresnet_block / mobile_bottleneck, bf16 rounding at the
device's storage points) and the native executor's block (vlsfr_iresnet_forward_blocks / _backward_blocks,
vlsfr_mobilenet_forward_units / _backward_units: the very kernels and context slots of the full pass) get the same
bf16-rounded input activation and output gradient; outputs, the input gradient, every parameter gradient and the running
statistics are compared per tensor.  One block does not amplify rounding noise, so residual / downsample / PReLU /
BatchNorm gradient routing errors cannot hide: tolerance 2e-2 of tensor scale (rel-L2), against 0.1 - 0.5 per tensor in
the 49-block band.  Reference blocks: model/resnet_arcface.py:26-55, model/mobilefacenet_def.py:27-52."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import backbones_ref as bb
from tests.golden import common

pytestmark = pytest.mark.gpu

TOL = 2e-2


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _bf16_round(t):
    return t.float().to(torch.bfloat16)


def _activation(rng, B, C, H, scale, offset):
    """An NCHW activation with per-channel means and spreads (so BatchNorm has something to normalise), bf16-exact."""
    x = rng.standard_normal((B, C, H, H)).astype(np.float32) * (scale * (0.5 + rng.random((1, C, 1, 1)).astype(np.float32)))
    x += offset * rng.standard_normal((1, C, 1, 1)).astype(np.float32)
    return _bf16_round(torch.from_numpy(x))


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


class _Native(object):
    """Raw access to a NativeBackbone's executor: handle, parameter / gradient / running tables, context and scratch."""

    def __init__(self, net, B):
        from vlsfr_amd import _lib
        from vlsfr_amd.model._native import _ptr_array, _stream
        self.L, self.net, self._ptr_array, self._stream = _lib.lib(), net, _ptr_array, _stream
        self.check = _lib.check
        dev = next(net.parameters()).device
        self.h, sizes = net._handle(B, dev)
        self.params, self.running = net._tables()
        net._prepare(self.h, sizes, self.params, dev)
        self.ctx = torch.zeros(sizes[1], dtype=torch.uint8, device=dev)
        self.scratch = net._scratch
        self.wcache = net._wcache
        self.grads = net._ensure_grads()

    def info(self, fn, k):
        out = (ctypes.c_int32 * 8)()
        f = getattr(self.L, fn)
        f.restype = ctypes.c_int
        self.check(f(self.h, ctypes.c_int32(k), out), fn)
        return list(out)

    def run(self, fn, k0, k1, inp, out, backward):
        f = getattr(self.L, fn)
        f.restype = ctypes.c_int
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        tab2 = self._ptr_array(self.grads) if backward else self._ptr_array(self.running)
        self.check(f(self.h, ctypes.c_int32(k0), ctypes.c_int32(k1), P(inp), self._ptr_array(self.params), tab2, P(self.wcache),
                     P(self.ctx), P(self.scratch), P(out), self._stream()), fn)


def _compare_block(name, net, nat, fwd, bwd, k0, k1, x, dout, oracle_block, prefix, tol=TOL):
    """x, dout: bf16-exact NCHW float tensors (CPU).  oracle_block(sd, x64) -> output (float64, emulated bf16 storage)."""
    sd = {k: (v.detach().double().cpu().clone() if v.is_floating_point() else v.detach().cpu().clone())
          for k, v in net.state_dict().items()}
    for k, v in sd.items():
        if k.startswith(prefix) and v.is_floating_point() and not bb.is_buffer(k):
            v.requires_grad_(True)
    xo = x.double().clone().requires_grad_(True)
    want = oracle_block(sd, xo)
    want.backward(dout.double())
    # native
    xin = _nhwc(x).to(torch.bfloat16).cuda()
    out = torch.empty(_nhwc(want.detach()).shape, dtype=torch.bfloat16, device="cuda")
    nat.run(fwd, k0, k1, xin, out, backward=False)
    dx = torch.empty_like(xin)
    for g in nat.grads:
        if g is not None:
            g.zero_()
    nat.run(bwd, k0, k1, _nhwc(dout).to(torch.bfloat16).cuda(), dx, backward=True)
    torch.cuda.synchronize()
    errs = {"out": rel_l2(_nchw(out.float().cpu()), want.detach()), "dX": rel_l2(_nchw(dx.float().cpu()), xo.grad)}
    named = dict(net.named_parameters())
    n_checked = 0
    for k, v in sd.items():
        if not (k.startswith(prefix) and v.is_floating_point()):
            continue
        if bb.is_buffer(k):
            if not k.endswith("num_batches_tracked"):
                got = dict(net.named_buffers())[k].detach().cpu().double()
                np.testing.assert_allclose(got.numpy(), v.detach().numpy(), rtol=5e-3, atol=5e-4, err_msg="%s %s" % (name, k))
            continue
        ref = v.grad
        got = named[k].grad.detach().cpu().double()
        if float(ref.abs().max()) < 1e-9:
            continue
        errs["d " + k] = rel_l2(got, ref)
        n_checked += 1
    worst = max(errs, key=errs.get)
    print("%s: %d parameter gradients; worst %s %.2e; out %.2e, dX %.2e" % (name, n_checked, worst, errs[worst], errs["out"], errs["dX"]))
    assert n_checked >= 8
    bad = {k: e for k, e in errs.items() if not e <= tol}
    assert not bad, bad


# (layers, block index, what): stride-2 blocks with the 1x1 + BN shortcut and stride-1 identity blocks, at the 64-channel
# 112/56 stage and at the 256-channel 14 x 14 stage that holds 55 % of ir100's FLOPs
IR_CASES = [((2, 1, 1, 1), 0, "layer1.0 stride 2 + downsample, 64 ch"), ((2, 1, 1, 1), 1, "layer1.1 stride 1, 64 ch, 56x56"),
            ((1, 1, 2, 1), 2, "layer3.0 stride 2 + downsample, 128 -> 256"), ((1, 1, 2, 1), 3, "layer3.1 stride 1, 256 ch, 14x14"),
            ((1, 1, 1, 2), 4, "layer4.1 stride 1, 512 ch, 7x7")]


@pytest.mark.parametrize("layers,k,what", IR_CASES, ids=[c[2].split(",")[0].replace(" ", "-") for c in IR_CASES])
def test_iresnet_block_teacher_forced(layers, k, what):
    from vlsfr_amd.model.iresnet import IResNet
    B, D = 16, 64
    sd0 = common.fill_state(bb.iresnet_state(layers, D), 17)
    net = IResNet(list(layers), feat_dim=D)
    net.load_state_dict(sd0)
    net = net.cuda().train()
    nat = _Native(net, B)
    cin, planes, H, Ho, stride, has_ds, _, nblk = nat.info("vlsfr_iresnet_block_info", k)
    assert nblk == sum(layers) and (stride == 2) == bool(has_ds)
    names, li, bi = [], 1, 0
    for li_, n_ in enumerate(layers, start=1):
        names += ["layer%d.%d" % (li_, b_) for b_ in range(n_)]
    pre = names[k]
    rng = np.random.default_rng(100 + k)
    x = _activation(rng, B, cin, H, 0.6, 0.3).float()
    dout = _activation(rng, B, planes, Ho, 2e-2, 1e-2).float()
    _compare_block(what, net, nat, "vlsfr_iresnet_forward_blocks", "vlsfr_iresnet_backward_blocks", k, k + 1, x, dout,
                   lambda sd, xo: bb.iresnet_block(sd, xo, pre, stride == 2, bb.Emu(True)), pre + ".")


def test_iresnet_two_blocks_chain_the_bn3_reduction():
    """Blocks layer3.0 + layer3.1 in one range: the input gradient of block k + 1 is written by the kernel that also
    accumulates the reduction of block k's bn3 (vlsfr_bn_backward_chain) — the one piece of wiring a single block cannot
    exercise."""
    from vlsfr_amd.model.iresnet import IResNet
    layers, B, D = (1, 1, 2, 1), 16, 64
    net = IResNet(list(layers), feat_dim=D)
    net.load_state_dict(common.fill_state(bb.iresnet_state(layers, D), 19))
    net = net.cuda().train()
    nat = _Native(net, B)
    cin, _, H, _, _, _, _, _ = nat.info("vlsfr_iresnet_block_info", 2)
    _, planes, _, Ho, _, _, _, _ = nat.info("vlsfr_iresnet_block_info", 3)
    rng = np.random.default_rng(7)
    x = _activation(rng, B, cin, H, 0.6, 0.3).float()
    dout = _activation(rng, B, planes, Ho, 2e-2, 1e-2).float()
    E = bb.Emu(True)
    _compare_block("layer3.0 + layer3.1", net, nat, "vlsfr_iresnet_forward_blocks", "vlsfr_iresnet_backward_blocks", 2, 4, x, dout,
                   lambda sd, xo: bb.iresnet_block(sd, bb.iresnet_block(sd, xo, "layer3.0", True, E), "layer3.1", False, E), "layer3.",
                   tol=2 * TOL)         # two blocks: measured 2.2e-2 on the BatchNorm biases (single blocks: <= 1.1e-2)


# BottleNeck bi = units 2 + 3 bi .. 4 + 3 bi (conv1 and dw_conv1 are units 0, 1)
MB_CASES = [(1, "blocks.1 connect (residual), 64 ch 28x28"), (0, "blocks.0 stride 2, no residual"),
            (5, "blocks.5 stride 2, 64 -> 128, t = 4"), (7, "blocks.7 connect, 128 ch 14x14")]


@pytest.mark.parametrize("bi,what", MB_CASES, ids=[c[1].split(",")[0].replace(" ", "-") for c in MB_CASES])
def test_mobilefacenet_bottleneck_teacher_forced(bi, what):
    from vlsfr_amd.model.mobilefacenet import MobileFaceNet, BOTTLENECKS
    B, D = 16, 128
    net = MobileFaceNet(feat_dim=D)
    net.load_state_dict(common.fill_state(bb.mobilefacenet_state(D), 23))
    net = net.cuda().train()
    nat = _Native(net, B)
    u0 = 2 + 3 * bi
    kind0, cin, mid, H, _, _, _, _ = nat.info("vlsfr_mobilenet_unit_info", u0)
    kind2, _, cout, _, Ho, res, _, _ = nat.info("vlsfr_mobilenet_unit_info", u0 + 2)
    assert (kind0, nat.info("vlsfr_mobilenet_unit_info", u0 + 1)[0], kind2) == (1, 2, 1)
    table = [(c, s if i == 0 else 1) for t, c, n, s in BOTTLENECKS for i in range(n)]
    stride = table[bi][1]
    connect = res >= 0
    assert connect == (stride == 1 and cin == cout) and (not connect or res == u0 - 1)
    rng = np.random.default_rng(200 + bi)
    x = _activation(rng, B, cin, H, 0.6, 0.3).float()
    dout = _activation(rng, B, cout, Ho, 2e-2, 1e-2).float()
    pre = "blocks.%d.conv" % bi
    _compare_block(what, net, nat, "vlsfr_mobilenet_forward_units", "vlsfr_mobilenet_backward_units", u0, u0 + 3, x, dout,
                   lambda sd, xo: bb.mobile_bottleneck(sd, xo, pre, stride, connect, bb.Emu(True)), pre + ".")
