"""The fused parameter sweeps (csrc/optim.hip through optim/fused.py) pinned directly: three consecutive SGD steps
from random parameters, gradients AND momentum buffers — so the momentum recursion, the weight decay and the nesterov
look-ahead are all visible — against the oracle's update rule (reference optim/optimizer.py:148-150 ->
torch.optim.SGD) and against torch.optim.SGD itself on the CPU; and the gallery EMA (reference ffc.py:139-145)."""
import numpy as np
import pytest
import torch

from oracle import ffc_ref

pytestmark = pytest.mark.gpu

# conv weight (channels_last memory, as the backbones keep them), a tensor longer than one 65536-element chunk,
# sizes that are not multiples of 4 (scalar tail of the kernel + padded views of the flat gradient buffer)
SHAPES = [(64, 32, 3, 3), (70001,), (513,), (7,), (256, 25), (1,)]


def _make(seed, device):
    g = torch.Generator().manual_seed(seed)
    ps = []
    for sh in SHAPES:
        t = torch.randn(sh, generator=g)
        if len(sh) == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        ps.append(t.to(device).requires_grad_(True))
    return ps


@pytest.mark.parametrize("nesterov", [True, False])
def test_fused_sgd_three_steps_match_the_update_rule(nesterov):
    from vlsfr_amd.optim.fused import FusedSGD
    lr, mu, wd = 0.1, 0.9, 1e-4
    dev = _make(0, "cuda")
    ref = [p.detach().cpu().double().clone() for p in dev]
    tor = [p.detach().cpu().clone().requires_grad_(True) for p in dev]
    opt = FusedSGD(dev, lr, momentum=mu, weight_decay=wd, nesterov=nesterov)
    topt = torch.optim.SGD(tor, lr, momentum=mu, weight_decay=wd, nesterov=nesterov)
    bufs = [None] * len(ref)
    gen = torch.Generator().manual_seed(1)
    for step in range(3):
        opt.zero_grad()
        grads = [torch.randn(p.shape, generator=gen) for p in ref]
        for p, t, g in zip(dev, tor, grads):
            if p.grad is None:
                p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
            p.grad.copy_(g.to(p.device))            # keeps whatever view / layout the optimizer attached
            t.grad = g.clone()
        bufs = ffc_ref.sgd_nesterov_step_ref(ref, [g.double() for g in grads], bufs, lr, mu, wd, nesterov=nesterov)
        opt.step()
        topt.step()
        torch.cuda.synchronize()
        for p, r, t in zip(dev, ref, tor):
            np.testing.assert_allclose(p.detach().cpu().numpy(), r.float().numpy(), rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(p.detach().cpu().numpy(), t.detach().numpy(), rtol=1e-6, atol=1e-6)
        for p, b in zip(dev, bufs):
            np.testing.assert_allclose(opt.state[p]["momentum_buffer"].cpu().numpy(), b.float().numpy(), rtol=1e-6, atol=1e-6)


def test_flat_gradient_views_are_16_byte_aligned():
    from vlsfr_amd.optim.fused import FusedSGD
    dev = _make(2, "cuda")
    opt = FusedSGD(dev, 0.1, momentum=0.9)
    flat = opt.flat_grad()
    assert flat is not None
    for p in dev:
        assert p.grad.data_ptr() % 16 == 0 and p.grad.shape == p.shape and p.grad.stride() == p.stride()
    flat.fill_(1.0)
    assert all(float(p.grad.min()) == 1.0 for p in dev)
    opt.zero_grad()
    assert float(flat.abs().max()) == 0.0


def test_ema_matches_reference_rule():
    from vlsfr_amd.optim.fused import ema_update
    gal, pro = _make(3, "cuda"), _make(4, "cuda")
    want = [g.detach().cpu().double() * 0.99 + p.detach().cpu().double() * (1.0 - 0.99) for g, p in zip(gal, pro)]   # ffc.py:144
    with torch.no_grad():
        ema_update(gal, pro, 0.99)
    torch.cuda.synchronize()
    for g, w in zip(gal, want):
        np.testing.assert_allclose(g.detach().cpu().numpy(), w.float().numpy(), rtol=1e-6, atol=1e-7)
