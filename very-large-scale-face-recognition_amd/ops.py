"""Thin ctypes wrappers over the per-operator device entry points of libvlsfr.so (sections 5-6 of
include/vlsfr.h).  Used by the per-kernel parity tests; the training path drives the same kernels
through the native backbone executor (backbone.py / csrc/iresnet.cpp) instead of one Python call
per operator."""
import ctypes

import torch

from . import _lib


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("N", "H", "W", "Cin", "Cout", "R", "S", "stride", "pad")]


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _call(name, *args):
    fn = getattr(_lib.lib(), name)
    fn.restype = ctypes.c_int
    _lib.check(fn(*args), name)


def out_hw(h, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1


def cast_weight(w_f32, rows, taps, C, Kp=None, transpose=True):
    """w_f32: fp32 device tensor whose memory is [rows][taps][C].  Returns (w bf16 [rows, Kp], wT bf16
    [C, taps, rows] or None)."""
    Kp = Kp or taps * C
    wb = torch.empty(rows, Kp, dtype=torch.bfloat16, device=w_f32.device)
    wT = torch.empty(C, taps, rows, dtype=torch.bfloat16, device=w_f32.device) if transpose else None
    _call("vlsfr_cast_weight", _p(w_f32), _p(wb), _p(wT), ctypes.c_int32(rows), ctypes.c_int32(taps),
          ctypes.c_int32(C), ctypes.c_int32(Kp), _st())
    return wb, wT


BN_REPL = int(_lib.lib().vlsfr_bn_repl())   # VLSFR_BN_REPL of the loaded library


def conv2d_fwd(x, w, desc, splitk=1, out_f32=False, stats=None):
    Ho, Wo = out_hw(desc.H, desc.R, desc.stride, desc.pad), out_hw(desc.W, desc.S, desc.stride, desc.pad)
    if out_f32:
        y = torch.zeros(desc.N, Ho, Wo, desc.Cout, dtype=torch.float32, device=x.device)
    else:
        y = torch.empty(desc.N, Ho, Wo, desc.Cout, dtype=torch.bfloat16, device=x.device)
    _call("vlsfr_conv2d_fwd", ctypes.byref(desc), _p(x), _p(w), _p(y), ctypes.c_int32(splitk),
          ctypes.c_int32(int(out_f32)), _p(stats), _st())
    return y


class BnIn(ctypes.Structure):      # vlsfr_bn_in
    _fields_ = [(n, ctypes.c_void_p) for n in ("scale", "shift", "slope", "a_out")]


def conv2d_fwd_bnin_supported(desc):
    L = _lib.lib()
    L.vlsfr_conv2d_fwd_bnin_supported.restype = ctypes.c_int32
    return bool(L.vlsfr_conv2d_fwd_bnin_supported(ctypes.byref(desc)))


def conv2d_fwd_bnin(x, w, desc, scale, shift, slope=None, want_a=True, stats=None):
    """y = conv(prelu(x * scale + shift)) with the BatchNorm / PReLU applied in the kernel's operand path (vlsfr_conv2d_fwd_bnin).
    Returns (y, a) with a = the transformed input (bf16) or None."""
    y = torch.empty(desc.N, desc.H, desc.W, desc.Cout, dtype=torch.bfloat16, device=x.device)
    a = torch.zeros(desc.N, desc.H, desc.W, desc.Cin, dtype=torch.bfloat16, device=x.device) if want_a else None
    ptr = lambda t: t.data_ptr() if t is not None else None
    b = BnIn(ptr(scale), ptr(shift), ptr(slope), ptr(a))
    _call("vlsfr_conv2d_fwd_bnin", ctypes.byref(desc), _p(x), _p(w), _p(y), _p(stats), ctypes.byref(b), _st())
    return y, a


def bn_finalize(sums, M, C, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    dev = sums.device
    mean, invstd, scale, shift = (torch.empty(C, dtype=torch.float32, device=dev) for _ in range(4))
    _call("vlsfr_bn_finalize", _p(sums), ctypes.c_int64(M), ctypes.c_int32(C), _p(gamma), _p(beta), ctypes.c_float(eps),
          ctypes.c_float(momentum), _p(mean), _p(invstd), _p(scale), _p(shift), _p(running_mean), _p(running_var), _st())
    return mean, invstd, scale, shift


def conv2d_dgrad(dy, wT, desc):
    dx = torch.empty(desc.N, desc.H, desc.W, desc.Cin, dtype=torch.bfloat16, device=dy.device)
    _call("vlsfr_conv2d_dgrad", ctypes.byref(desc), _p(dy), _p(wT), _p(dx), _st())
    return dx


def conv2d_wgrad(dy, x, desc, dw=None, splitk=0):
    if dw is None:
        dw = torch.zeros(desc.Cout, desc.R, desc.S, desc.Cin, dtype=torch.float32, device=x.device)
    _call("vlsfr_conv2d_wgrad", ctypes.byref(desc), _p(dy), _p(x), _p(dw), ctypes.c_int32(splitk), _st())
    return dw


def new_sums(C, device):
    return torch.zeros(BN_REPL, 2, C, dtype=torch.float64, device=device)       # float64 sums (include/vlsfr.h section 6)


def bn_stats(x, M, C):
    sums = new_sums(C, x.device)
    _call("vlsfr_bn_stats", _p(x), ctypes.c_int64(M), ctypes.c_int32(C), _p(sums), _st())
    return sums


def bn_apply(x, M, C, HW, sums, gamma, beta, slope=None, residual=None, running_mean=None, running_var=None,
             out_nchw=False, eps=1e-5, momentum=0.1, out_sums=None):
    y = torch.empty(M * C, dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    _call("vlsfr_bn_apply", _p(x), _p(y), ctypes.c_int64(M), ctypes.c_int32(C), ctypes.c_int32(HW), _p(sums),
          _p(gamma), _p(beta), _p(slope), _p(residual), _p(mean), _p(invstd), _p(running_mean), _p(running_var),
          ctypes.c_float(eps), ctypes.c_float(momentum), _p(out_sums), ctypes.c_int32(int(out_nchw)), _st())
    return y, mean, invstd


def bn_backward(dy, x, M, C, HW, mean, invstd, gamma, beta, slope=None, dx_add=None, dgamma=None, dbeta=None,
                dslope=None, dy_nchw=False):
    dx = torch.empty(M * C, dtype=torch.bfloat16, device=x.device)
    red = torch.zeros(BN_REPL, 3, C, dtype=torch.float32, device=x.device)
    _call("vlsfr_bn_backward", _p(dy), _p(x), _p(dx), ctypes.c_int64(M), ctypes.c_int32(C), ctypes.c_int32(HW),
          _p(mean), _p(invstd), _p(gamma), _p(beta), _p(slope), _p(red), _p(dx_add), _p(dgamma), _p(dbeta),
          _p(dslope), ctypes.c_int32(int(dy_nchw)), _st())
    return dx


def stem_im2col(x_nchw, stride=1):
    N, _, H, W = x_nchw.shape
    Ho, Wo = out_hw(H, 3, stride, 1), out_hw(W, 3, stride, 1)
    out = torch.empty(N * Ho * Wo, 32, dtype=torch.bfloat16, device=x_nchw.device)
    _call("vlsfr_stem_im2col", _p(x_nchw), _p(out), ctypes.c_int32(N), ctypes.c_int32(H), ctypes.c_int32(W),
          ctypes.c_int32(stride), _st())
    return out


def dwconv_fwd(x, w, desc, stats=None):
    Ho, Wo = out_hw(desc.H, desc.R, desc.stride, desc.pad), out_hw(desc.W, desc.S, desc.stride, desc.pad)
    y = torch.empty(desc.N, Ho, Wo, desc.Cout, dtype=torch.bfloat16, device=x.device)
    _call("vlsfr_dwconv_fwd", ctypes.byref(desc), _p(x), _p(w), _p(y), _p(stats), _st())
    return y


def dwconv_dgrad(dy, w, desc):
    dx = torch.empty(desc.N, desc.H, desc.W, desc.Cin, dtype=torch.bfloat16, device=dy.device)
    _call("vlsfr_dwconv_dgrad", ctypes.byref(desc), _p(dy), _p(w), _p(dx), _st())
    return dx


def dwconv_wgrad(dy, x, desc, dw=None):
    if dw is None:
        dw = torch.zeros(desc.Cout, desc.R * desc.S, dtype=torch.float32, device=x.device)
    _call("vlsfr_dwconv_wgrad", ctypes.byref(desc), _p(dy), _p(x), _p(dw), _st())
    return dw


def dwconv_wgrad_ws(dy, x, desc, dw=None):
    """The depthwise weight gradient through per-block partial sums in a workspace (vlsfr_dwconv_wgrad_ws)."""
    if dw is None:
        dw = torch.zeros(desc.Cout, desc.R * desc.S, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    L.vlsfr_dwconv_wgrad_workspace_bytes.restype = ctypes.c_size_t
    n = int(L.vlsfr_dwconv_wgrad_workspace_bytes(ctypes.byref(desc)))
    ws = torch.empty(max(n, 16), dtype=torch.uint8, device=x.device)
    _call("vlsfr_dwconv_wgrad_ws", ctypes.byref(desc), _p(dy), _p(x), _p(dw), _p(ws), ctypes.c_size_t(n), _st())
    return dw


def embed_fwd(fc, fc_bias, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    B, D = fc.shape
    mk = lambda *s: torch.empty(*s, dtype=torch.float32, device=fc.device)
    z, xhat, invstd, emb, inv_norm = mk(B, D), mk(B, D), mk(D), mk(B, D), mk(B)
    _call("vlsfr_embed_fwd", _p(fc), _p(fc_bias), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(z),
          _p(xhat), _p(invstd), _p(emb), _p(inv_norm), ctypes.c_int32(B), ctypes.c_int32(D), ctypes.c_float(eps),
          ctypes.c_float(momentum), _st())
    return emb, (z, xhat, invstd, inv_norm)


def embed_bwd(demb, emb, saved, gamma, dbeta, dfc_bias, dgamma=None):
    z, xhat, invstd, inv_norm = saved
    B, D = emb.shape
    dz = torch.empty(B, D, dtype=torch.float32, device=emb.device)
    dfc = torch.empty(B, D, dtype=torch.bfloat16, device=emb.device)
    _call("vlsfr_embed_bwd", _p(demb), _p(emb), _p(inv_norm), _p(xhat), _p(invstd), _p(gamma), _p(dz), _p(dfc),
          _p(dbeta), _p(dfc_bias), _p(dgamma), ctypes.c_int32(B), ctypes.c_int32(D), _st())
    return dfc


# ---- torchvision-style ResNet operators (csrc/resnet_ops.hip; include/vlsfr.h section 7c) -------------------
def stem7_im2col(x_nchw):
    N, _, H, W = x_nchw.shape
    Ho, Wo = out_hw(H, 7, 2, 3), out_hw(W, 7, 2, 3)
    out = torch.empty(N * Ho * Wo, 160, dtype=torch.bfloat16, device=x_nchw.device)
    _call("vlsfr_stem7_im2col", _p(x_nchw), _p(out), ctypes.c_int32(N), ctypes.c_int32(H), ctypes.c_int32(W), _st())
    return out


def maxpool_fwd(x):
    N, H, W, C = x.shape
    y = torch.empty(N, out_hw(H, 3, 2, 1), out_hw(W, 3, 2, 1), C, dtype=torch.bfloat16, device=x.device)
    _call("vlsfr_maxpool3x3s2_fwd", _p(x), _p(y), ctypes.c_int32(N), ctypes.c_int32(H), ctypes.c_int32(W), ctypes.c_int32(C), _st())
    return y


def maxpool_bwd(dy, x, y):
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    _call("vlsfr_maxpool3x3s2_bwd", _p(dy), _p(x), _p(y), _p(dx), ctypes.c_int32(N), ctypes.c_int32(H), ctypes.c_int32(W),
          ctypes.c_int32(C), _st())
    return dx


def relu_bwd(dy, y, M, C, HW, in_nchw=False):
    dx = torch.empty(M * C, dtype=torch.bfloat16, device=dy.device)
    _call("vlsfr_relu_bwd_bf16", _p(dy), _p(y), _p(dx), ctypes.c_int64(M), ctypes.c_int32(C), ctypes.c_int32(HW),
          ctypes.c_int32(int(in_nchw)), _st())
    return dx


def conv2d_wgrad_ws(dy, x, desc, dw=None, splitk=0):
    """Weight gradient through the slab path (split-K slices to a workspace, ordered reduction)."""
    if dw is None:
        dw = torch.zeros(desc.Cout, desc.R, desc.S, desc.Cin, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    L.vlsfr_conv2d_wgrad_workspace_bytes.restype = ctypes.c_size_t
    n = L.vlsfr_conv2d_wgrad_workspace_bytes(ctypes.byref(desc), ctypes.c_int32(splitk))
    ws = torch.empty(max(n, 16), dtype=torch.uint8, device=x.device)
    _call("vlsfr_conv2d_wgrad_ws", ctypes.byref(desc), _p(dy), _p(x), _p(dw), ctypes.c_int32(splitk), _p(ws),
          ctypes.c_size_t(n), _st())
    return dw, n


def conv2d_wgrad_group(dys, xs, desc, dws=None, splitk=0, slabs=False):
    """Weight gradients of len(dys) <= 4 layers of one descriptor in a single launch (vlsfr_conv2d_wgrad_group)."""
    n = len(dys)
    if dws is None:
        dws = [torch.zeros(desc.Cout, desc.R, desc.S, desc.Cin, dtype=torch.float32, device=xs[0].device) for _ in range(n)]
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    L = _lib.lib()
    ws, nb = None, 0
    if slabs:
        L.vlsfr_conv2d_wgrad_group_workspace_bytes.restype = ctypes.c_size_t
        nb = L.vlsfr_conv2d_wgrad_group_workspace_bytes(ctypes.byref(desc), ctypes.c_int32(n), ctypes.c_int32(splitk))
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=xs[0].device)
    _call("vlsfr_conv2d_wgrad_group", ctypes.byref(desc), ctypes.c_int32(n), arr(dys), arr(xs), arr(dws), ctypes.c_int32(splitk),
          _p(ws), ctypes.c_size_t(nb), _st())
    return dws


class BnRed(ctypes.Structure):      # vlsfr_bn_red
    _fields_ = [(n, ctypes.c_void_p) for n in ("x", "mean", "invstd", "gamma", "beta", "slope", "red")]


def conv2d_dgrad_bnred(dy, wT, desc, x, mean, invstd, gamma=None, beta=None, slope=None):
    """Input gradient with the reduction of the BatchNorm (+ PReLU) backward that consumes it accumulated by the epilogue
    (vlsfr_conv2d_dgrad_bnred).  Returns (dx, red [BN_REPL, 3, Cin])."""
    dx = torch.empty(desc.N, desc.H, desc.W, desc.Cin, dtype=torch.bfloat16, device=dy.device)
    red = torch.zeros(BN_REPL, 3, desc.Cin, dtype=torch.float32, device=dy.device)
    ptr = lambda t: t.data_ptr() if t is not None else None
    r = BnRed(ptr(x), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(slope), ptr(red))
    _call("vlsfr_conv2d_dgrad_bnred", ctypes.byref(desc), _p(dy), _p(wT), _p(dx), ctypes.byref(r), _st())
    return dx, red


def bn_backward_reduce(dy, x, M, C, HW, mean, invstd, gamma=None, beta=None, slope=None):
    red = torch.zeros(BN_REPL, 3, C, dtype=torch.float32, device=x.device)
    _call("vlsfr_bn_backward_reduce", _p(dy), _p(x), ctypes.c_int64(M), ctypes.c_int32(C), ctypes.c_int32(HW), _p(mean),
          _p(invstd), _p(gamma), _p(beta), _p(slope), _p(red), _st())
    return red
