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


class PartitionedSGD(FusedSGD):
    """SGD-nesterov with the optimizer state and the update partitioned over the ranks of a node (ZeRO-1; the
    "partitioned SGD" BASELINE.json's north_star adds to the reference's torch.optim.SGD, optim/optimizer.py:148-150).

    Parameters and gradients live in two flat fp32 buffers with one layout: buckets (groups of parameters whose
    gradients complete together in the backward pass, backward order), every parameter view 16-byte aligned, every
    bucket padded to a multiple of 4 * world elements.  Per bucket and step:
        reduce_bucket(b)   reduce-scatter(sum) of the bucket's gradients -> this rank's 1/world slice   [RCCL]
        step()             fused update (csrc/optim.hip) of the rank's parameter slice against its momentum slice,
                           then all-gather of the updated slices into every rank's flat parameter buffer     [RCCL]
    Elementwise the update is torch.optim.SGD's, so the result equals the replicated FusedSGD on summed gradients
    (tests/test_parallel_cpu.py runs both over gloo).  Momentum memory is 1/world per rank."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, nesterov=False, comm=None, bucket_of=None,
                 n_buckets=1):
        params = list(params)
        super(PartitionedSGD, self).__init__(params, lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov)
        self.comm = comm
        self.world, self.rank = comm.world, comm.rank
        self.n_buckets = int(n_buckets)
        self._bucket_of = bucket_of or (lambda p: 0)
        self._part = None

    # ------------------------------------------------------------------------------------------
    def _params(self):
        return [p for g in self.param_groups for p in g["params"] if p.requires_grad]

    def partition(self):
        """Builds the flat buffers (once, when the parameters are on their device) and re-points p.data / p.grad."""
        if self._part is not None:
            return self._part
        ps = self._params()
        dev = ps[0].device
        pad4 = lambda n: (n + 3) & ~3
        unit = 4 * self.world
        buckets = [[] for _ in range(self.n_buckets)]
        for p in ps:
            if p.dtype != torch.float32 or not _dense_same_layout([p.data]):
                raise _lib.VlsfrError("PartitionedSGD: parameters must be dense fp32 tensors")
            buckets[self._bucket_of(p)].append(p)
        ranges, off = [], 0
        for bp in buckets:
            n = sum(pad4(p.numel()) for p in bp)
            n = (n + unit - 1) // unit * unit
            ranges.append((off, n))
            off += n
        flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        for (start, _), bp in zip(ranges, buckets):
            o = start
            for p in bp:
                vp = torch.as_strided(flat_p, p.shape, p.stride(), storage_offset=o)
                vg = torch.as_strided(flat_g, p.shape, p.stride(), storage_offset=o)
                vp.copy_(p.data)
                if p.grad is not None:
                    vg.copy_(p.grad)
                p.data = vp
                p.grad = vg
                owner = getattr(p, "_vlsfr_owner", None)
                if owner is not None:
                    owner.weights_dirty = True
                o += pad4(p.numel())
        shard = lambda buf, start, n: buf[start + self.rank * (n // self.world):start + (self.rank + 1) * (n // self.world)]
        self._part = dict(
            buckets=buckets, ranges=ranges, flat_p=flat_p, flat_g=flat_g,
            pshard=[shard(flat_p, s, n).clone() for s, n in ranges],       # fp32 master slices of this rank
            gshard=[torch.zeros(n // self.world, dtype=torch.float32, device=dev) for _, n in ranges],
            mshard=[torch.zeros(n // self.world, dtype=torch.float32, device=dev) for _, n in ranges])
        self._flat_grad = flat_g
        self._flat_key = tuple(p.grad.data_ptr() for p in ps)
        return self._part

    def flat_grad(self):
        return self.partition()["flat_g"]

    def zero_grad(self, set_to_none=False):
        self.partition()["flat_g"].zero_()

    # ------------------------------------------------------------------------------------------
    def reduce_bucket(self, b):
        """Sum of bucket b's gradients over the ranks, this rank's slice only (on the current stream)."""
        part = self.partition()
        start, n = part["ranges"][b]
        self.comm.reduce_scatter_sum(part["gshard"][b], part["flat_g"][start:start + n])

    def _update_shards(self, group):
        part = self._part
        groups = [(part["pshard"][b], part["gshard"][b], part["mshard"][b]) for b in range(self.n_buckets)
                  if part["pshard"][b].numel()]
        cache = self._caches.setdefault("shards", _TableCache())
        tab = cache.get(groups, part["flat_p"].device)
        fn = _lib.lib().vlsfr_sgd_nesterov
        fn.restype = ctypes.c_int
        _lib.check(fn(ctypes.c_void_p(tab.data_ptr()), ctypes.c_int32(tab.shape[0]), ctypes.c_float(group["lr"]),
                      ctypes.c_float(group["momentum"]), ctypes.c_float(group["weight_decay"]),
                      ctypes.c_int32(int(group["nesterov"])),
                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "vlsfr_sgd_nesterov")

    @torch.no_grad()
    def step(self, closure=None):
        """Expects reduce_bucket(b) to have run for every bucket since the last backward pass."""
        part = self.partition()
        if len(self.param_groups) != 1:
            raise _lib.VlsfrError("PartitionedSGD: one parameter group")
        self._update_shards(self.param_groups[0])
        for b, (start, n) in enumerate(part["ranges"]):
            if n:
                self.comm.all_gather_into(part["flat_p"][start:start + n], part["pshard"][b])
        for p in self._params():
            owner = getattr(p, "_vlsfr_owner", None)
            if owner is not None:
                owner.weights_dirty = True
        return None

    # ------------------------------------------------------------------------------------------
    def consolidate_state(self):
        """All-gathers the momentum slices and exposes them as the per-parameter `momentum_buffer` entries torch's
        Optimizer.state_dict() serialises (collective: every rank calls it; used for checkpoints)."""
        part = self.partition()
        for b, ((start, n), bp) in enumerate(zip(part["ranges"], part["buckets"])):
            full = torch.empty(n, dtype=torch.float32, device=part["flat_p"].device)
            self.comm.all_gather_into(full, part["mshard"][b])
            o = 0
            for p in bp:
                self.state[p]["momentum_buffer"] = torch.as_strided(full, p.shape, p.stride(), storage_offset=o).clone()
                o += (p.numel() + 3) & ~3

    def release_consolidated(self):
        """Drops the full-size momentum_buffer entries consolidate_state() left in self.state (the step itself uses the
        1 / world slices): after a checkpoint has been written the momentum memory is 1 / world per rank again."""
        for st in self.state.values():
            st.pop("momentum_buffer", None)

    def scatter_state(self):
        """Inverse of consolidate_state (after load_state_dict): this rank's momentum slices from the full buffers."""
        part = self.partition()
        for b, ((start, n), bp) in enumerate(zip(part["ranges"], part["buckets"])):
            full = torch.zeros(n, dtype=torch.float32, device=part["flat_p"].device)
            o = 0
            for p in bp:
                mb = self.state.get(p, {}).get("momentum_buffer")
                if mb is not None:
                    torch.as_strided(full, p.shape, p.stride(), storage_offset=o).copy_(mb)
                o += (p.numel() + 3) & ~3
            k = n // self.world
            part["mshard"][b].copy_(full[self.rank * k:(self.rank + 1) * k])
