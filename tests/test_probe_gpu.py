"""Pins the gfx950 fragment conventions (MFMA 16x16x32 bf16 operand/accumulator maps, transposed
LDS read) with exact small-integer data.  Asymmetric operands, so a row/column swap cannot hide."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _call(fn, *args):
    from vlsfr_amd import _lib
    fn.restype = ctypes.c_int
    _lib.check(fn(*args), fn.__name__)


@pytest.mark.parametrize("rs", [32, 64, 1056, 288])
def test_mfma_with_transposed_lds_operand(rs):
    from vlsfr_amd import _lib
    rng = np.random.default_rng(0)
    A = rng.integers(-4, 5, size=(16, 32)).astype(np.float32)
    B = rng.integers(-4, 5, size=(32, 16)).astype(np.float32)
    a, b = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    c = torch.zeros(16, 16, device="cuda")
    _call(_lib.lib().vlsfr_probe_mfma_tr, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()),
          ctypes.c_void_p(c.data_ptr()), ctypes.c_int32(rs), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    np.testing.assert_array_equal(c.cpu().numpy(), A @ B)


def test_mfma_natural_k_order():
    from vlsfr_amd import _lib
    rng = np.random.default_rng(1)
    A = rng.integers(-4, 5, size=(16, 32)).astype(np.float32)
    Bt = rng.integers(-4, 5, size=(16, 32)).astype(np.float32)
    a, b = torch.from_numpy(A).cuda(), torch.from_numpy(Bt).cuda()
    c = torch.zeros(16, 16, device="cuda")
    _call(_lib.lib().vlsfr_probe_mfma_nat, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()),
          ctypes.c_void_p(c.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    np.testing.assert_array_equal(c.cpu().numpy(), A @ Bt.T)
