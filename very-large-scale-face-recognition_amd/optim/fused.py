"""Fused parameter sweeps: SGD-nesterov step and the gallery EMA, one kernel launch each over all
tensors (csrc/optim.hip).  Elementwise over storage, so a tensor and its partners (grad, momentum
buffer, EMA twin) must share one dense layout — which holds for tensors made with
``zeros_like(..., memory_format=preserve_format)`` / ``copy_``."""
import ctypes

import numpy as np
import torch

from .. import _lib

CHUNK = 65536


def _dense_same_layout(ts):
    s0 = ts[0].stride()
    for t in ts:
        if t.stride() != s0 or t.shape != ts[0].shape or t.dtype != torch.float32:
            return False
    t = ts[0]
    return t.is_contiguous() or (t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous())


def _chunk_table(groups):
    """groups: list of tuples of same-layout tensors.  Returns int64 [n_chunks, len(tuple) + 1]."""
    rows = []
    for ts in groups:
        if not _dense_same_layout(ts):
            raise _lib.VlsfrError("fused sweep: tensors of one group must be fp32, dense and share strides")
        n = ts[0].numel()
        if any(t.data_ptr() % 16 for t in ts):
            raise _lib.VlsfrError("fused sweep: tensor storage must be 16-byte aligned")
        for off in range(0, n, CHUNK):
            rows.append([t.data_ptr() + 4 * off for t in ts] + [min(CHUNK, n - off)])
    return np.asarray(rows, dtype=np.int64).reshape(len(rows), len(groups[0]) + 1 if groups else 1)


class _TableCache(object):
    def __init__(self):
        self.key, self.dev = None, None

    def get(self, groups, device):
        key = tuple(t.data_ptr() for ts in groups for t in ts)
        if key != self.key:
            tab = _chunk_table(groups)
            self.dev = torch.from_numpy(tab).to(device)
            self.key = key
        return self.dev


_ema_cache = {}


def ema_update(gallery_params, probe_params, m):
    """gallery <- m * gallery + (1 - m) * probe over every parameter (ffc.py:139-145)."""
    groups = [(g.data, p.data) for g, p in zip(gallery_params, probe_params)]
    if not groups:
        return
    dev = groups[0][0].device
    cache = _ema_cache.setdefault(id(gallery_params[0]), _TableCache())
    tab = cache.get(groups, dev)
    fn = _lib.lib().vlsfr_ema
    fn.restype = ctypes.c_int
    _lib.check(fn(ctypes.c_void_p(tab.data_ptr()), ctypes.c_int32(tab.shape[0]), ctypes.c_float(m),
                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "vlsfr_ema")


class FusedSGD(torch.optim.Optimizer):
    """Drop-in for torch.optim.SGD(params, lr, momentum, weight_decay, nesterov) on device tensors."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, nesterov=False, dampening=0.0):
        if dampening != 0.0:
            raise _lib.VlsfrError("FusedSGD: dampening is not covered (the reference uses 0)")
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov, dampening=0.0)
        super(FusedSGD, self).__init__(params, defaults)
        self._caches = {}
        self._flat_grad = None

    def _attach_flat_grads(self):
        """One flat fp32 gradient buffer; every p.grad becomes a view of it with p's own (dense)
        strides, so zero_grad is a single memset and a multi-GPU all-reduce needs no packing."""
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        if not ps or not ps[0].is_cuda:
            return None
        pad4 = lambda n: (n + 3) & ~3          # every view starts 16-byte aligned: the sweeps use 16-byte accesses
        total = sum(pad4(p.numel()) for p in ps)
        flat = torch.zeros(total, dtype=torch.float32, device=ps[0].device)
        off = 0
        for p in ps:
            if p.dtype != torch.float32 or not _dense_same_layout([p.data]):
                return None
            view = torch.as_strided(flat, p.shape, p.stride(), storage_offset=off)
            if p.grad is not None:
                view.copy_(p.grad)
            p.grad = view
            off += pad4(p.numel())             # the padding stays zero (harmless in an all-reduce)
        self._flat_grad = flat
        self._flat_key = tuple(p.grad.data_ptr() for p in ps)
        return flat

    def flat_grad(self):
        """The flat gradient buffer (attached on first use), or None if the layout does not allow it."""
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        if self._flat_grad is None or any(p.grad is None for p in ps) or \
                self._flat_key != tuple(p.grad.data_ptr() for p in ps):
            return self._attach_flat_grads()
        return self._flat_grad

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        fn = _lib.lib().vlsfr_sgd_nesterov
        fn.restype = ctypes.c_int
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            if not ps[0].is_cuda:
                raise _lib.VlsfrError("FusedSGD.step needs device parameters: there is no CPU path")
            groups = []
            for p in ps:
                st = self.state[p]
                if "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif st["momentum_buffer"].stride() != p.stride():      # e.g. after load_state_dict: the kernel walks raw memory
                    st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st["momentum_buffer"])
                groups.append((p.data, p.grad, st["momentum_buffer"]))
            cache = self._caches.setdefault(gi, _TableCache())
            tab = cache.get(groups, ps[0].device)
            _lib.check(fn(ctypes.c_void_p(tab.data_ptr()), ctypes.c_int32(tab.shape[0]), ctypes.c_float(group["lr"]),
                          ctypes.c_float(group["momentum"]), ctypes.c_float(group["weight_decay"]),
                          ctypes.c_int32(int(group["nesterov"])),
                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "vlsfr_sgd_nesterov")
            for p in ps:
                owner = getattr(p, "_vlsfr_owner", None)
                if owner is not None:
                    owner.weights_dirty = True
        return loss

    def zero_grad(self, set_to_none=False):
        """Gradients are accumulated in place by the backbone executor, so keep the buffers and
        clear them (set_to_none=True is honoured but costs a re-allocation on the next backward)."""
        if set_to_none:
            self._flat_grad = None
            return super(FusedSGD, self).zero_grad(set_to_none=True)
        flat = self.flat_grad()
        if flat is not None:
            flat.zero_()
            return
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    p.grad.zero_()
