"""Dynamic Class Pool head: host bookkeeping (native LRU) + the fused gfx950 head kernels.

One ``DcpHead.run_pass`` is one ``FFC.forward_impl`` (committing, reference ffc.py:153-204) or one
``FFC.forward_impl_rollback`` (transactional, ffc.py:208-260) *after* the two backbone calls: it
takes the probe/gallery embeddings and the two label vectors and returns the pass's loss as an
autograd node whose backward is the dL/dp the kernel already produced (the pool is a buffer, so
nothing else needs a gradient).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .lru import LRU

LOSS_TYPES = {"AM": 0, "Arc": 1, "SV": 2}


class HeadCfg(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("D", ctypes.c_int32), ("Q", ctypes.c_int64),
                ("loss_type", ctypes.c_int32), ("scale", ctypes.c_float), ("margin", ctypes.c_float),
                ("hard_neg", ctypes.c_int32), ("precise", ctypes.c_int32), ("n_chunks", ctypes.c_int32),
                ("slot_lo", ctypes.c_int32), ("n_rows_total", ctypes.c_int32), ("pool_bf16", ctypes.c_void_p),
                ("pool_fp8", ctypes.c_void_p)]


def _check_abi(L):
    """The ctypes mirror of vlsfr_head_cfg must be the struct libvlsfr.so was compiled with."""
    L.vlsfr_head_cfg_size.restype = ctypes.c_size_t
    n = L.vlsfr_head_cfg_size()
    if n != ctypes.sizeof(HeadCfg):
        raise _lib.VlsfrError("vlsfr_head_cfg: library has %d bytes, binding has %d — rebuild libvlsfr.so" %
                              (n, ctypes.sizeof(HeadCfg)))


class PoolShadow(object):
    """Reduced-precision mirror of queue[0] that the head sweep streams (include/vlsfr.h section 3): built on first use,
    kept current after vlsfr_pool_scatter, rebuilt when queue was modified through torch (its version counter moved:
    load_state_dict, copy_).  Only for D = 512 and plain operands — the cases csrc/head16.hip / head8.hip cover.
    dtype "bf16": [Q, 512] bf16, updated by vlsfr_pool_scatter itself; "fp8": the fragment-major e4m3 image of
    csrc/head8.hip (config C5's precision), updated by vlsfr_pool_shadow8_update after the scatter."""

    def __init__(self, queue):
        self.queue = queue
        self.t = {}            # dtype -> tensor
        self.version = {}

    def _usable(self, enabled):
        return enabled and self.queue.shape[2] == 512 and os.environ.get("VLSFR_HEAD_SHADOW", "1") != "0"

    def ptr(self, enabled, dtype="bf16"):
        """Device pointer of the current mirror in `dtype` (None when the sweep must read the fp32 pool)."""
        q = self.queue
        if not self._usable(enabled):
            return None
        L = _lib.lib()
        if dtype not in self.t:
            if dtype == "fp8":
                L.vlsfr_pool_shadow8_bytes.restype = ctypes.c_size_t
                self.t[dtype] = torch.empty(int(L.vlsfr_pool_shadow8_bytes(ctypes.c_int64(q.shape[1]))), dtype=torch.uint8, device=q.device)
            else:
                self.t[dtype] = torch.empty(q.shape[1], q.shape[2], dtype=torch.bfloat16, device=q.device)
        t = self.t[dtype]
        if self.version.get(dtype) != q._version:
            fn = L.vlsfr_pool_shadow8_build if dtype == "fp8" else L.vlsfr_pool_shadow_build
            fn.restype = ctypes.c_int
            _lib.check(fn(ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(t.data_ptr()), ctypes.c_int64(q.shape[1]),
                          ctypes.c_int32(q.shape[2]), _stream_ptr()), "vlsfr_pool_shadow_build")
            self.version[dtype] = q._version
        return t.data_ptr()

    def invalidate(self):
        """After a write to the pool that torch's version counter does not see (through `.data`, a raw pointer, an
        in-place collective on `.data`): the next ptr() rebuilds every mirror."""
        self.version.clear()

    def scatter_target(self):
        """The bf16 mirror's pointer for vlsfr_pool_scatter (which writes the fp32 row and its bf16 image in one launch)."""
        t = self.t.get("bf16")
        return self.ptr(True) if t is not None else None

    def after_scatter(self, cols_ptr, n, slot_lo=0):
        """Refresh the fp8 images of the slots vlsfr_pool_scatter just wrote (cols: device int32, global slot ids)."""
        t = self.t.get("fp8")
        if t is None or self.version.get("fp8") != self.queue._version:
            return                                   # not built yet, or stale anyway: the next ptr() rebuilds it
        q = self.queue
        fn = _lib.lib().vlsfr_pool_shadow8_update
        fn.restype = ctypes.c_int
        _lib.check(fn(ctypes.c_void_p(q.data_ptr()), ctypes.c_int64(q.shape[1]), ctypes.c_int32(q.shape[2]), cols_ptr,
                      ctypes.c_int32(n), ctypes.c_int32(slot_lo), ctypes.c_void_p(t.data_ptr()), _stream_ptr()),
                   "vlsfr_pool_shadow8_update")


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, loss, dP):
        ctx.save_for_backward(dP)
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out):
        (dP,) = ctx.saved_tensors
        return grad_out * dP, None, None


class QueuePositionView(object):
    """dict-like face of the reference's ``queue_position_dict`` (ffc.py:41-43) over the native
    uint8[Q] array."""

    def __init__(self, arr):
        self._a = arr

    def __len__(self):
        return self._a.shape[0]

    def __getitem__(self, i):
        return int(self._a[i])

    def __setitem__(self, i, v):
        self._a[i] = v

    def __iter__(self):
        return iter(range(self._a.shape[0]))

    def keys(self):
        return range(self._a.shape[0])

    def values(self):
        return self._a.tolist()

    def items(self):
        return enumerate(self._a.tolist())

    def __eq__(self, other):
        return dict(self.items()) == dict(other.items() if hasattr(other, "items") else other)

    def to_dict(self):
        return dict(self.items())


class DcpHead(object):
    def __init__(self, queue, scale, margin, loss_type, precise=False, n_chunks=0):
        assert loss_type in LOSS_TYPES
        assert queue.dim() == 3 and queue.shape[0] == 2 and queue.dtype == torch.float32
        self.L = _lib.lib()
        _check_abi(self.L)
        self.queue = queue                       # [2, Q, D] fp32, device, contiguous (shared with FFC.queue)
        self.shadow = PoolShadow(queue)
        self.Q, self.D = int(queue.shape[1]), int(queue.shape[2])
        self.scale, self.margin, self.loss_type = float(scale), float(margin), loss_type
        self.precise, self.n_chunks = bool(precise), int(n_chunks or os.environ.get('VLSFR_HEAD_CHUNKS', 0))
        self.head_dtype = os.environ.get('VLSFR_HEAD_DTYPE', 'bf16')   # "fp8": the e4m3 sweep (config C5's precision)
        self.hard_neg = min(max(int(self.Q * 0.0002), 3), 10)          # ffc.py:48
        self.lru = LRU(self.Q)                                        # ffc.py:40
        self.qp = np.zeros(self.Q, dtype=np.uint8)                    # ffc.py:41-43
        self._ws = None
        self._ws_key = None

    # -------------------------------------------------------------------------------------------
    def _cfg(self, B, B_total=0):
        fp8 = self.head_dtype == "fp8"
        return HeadCfg(B, self.D, self.Q, LOSS_TYPES[self.loss_type], self.scale, self.margin, self.hard_neg,
                       int(self.precise), self.n_chunks, 0, B_total, None if fp8 else self.shadow.ptr(not self.precise),
                       self.shadow.ptr(not self.precise, "fp8") if fp8 else None)

    def _workspace(self, cfg, device):
        key = (cfg.B, cfg.n_rows_total, str(device), cfg.pool_bf16, cfg.pool_fp8, cfg.precise, cfg.loss_type, cfg.scale, cfg.n_chunks)
        if self._ws_key != key:
            self.L.vlsfr_head_workspace_bytes.restype = ctypes.c_size_t
            self.L.vlsfr_head_workspace_bytes.argtypes = [ctypes.POINTER(HeadCfg)]
            n = self.L.vlsfr_head_workspace_bytes(ctypes.byref(cfg))
            if n == 0:
                raise _lib.VlsfrError("vlsfr_head_workspace_bytes: " + self.L.vlsfr_last_error().decode())
            self._ws = torch.empty(n, dtype=torch.uint8, device=device)
            self._ws_key = key
        return self._ws

    def assign(self, probe_label, gallery_label, transactional):
        """Host bookkeeping of one pass (vlsfr_dcp_assign)."""
        gl = np.ascontiguousarray(np.asarray(gallery_label, dtype=np.int64))
        pl = np.ascontiguousarray(np.asarray(probe_label, dtype=np.int64))
        n = gl.shape[0]
        assert pl.shape[0] == n
        # one packed int32 table: pool_label | special_col | src1 | src2 | rows | cols | ones
        tab = np.zeros(13 * max(n, 1), dtype=np.int32)
        o = lambda k: tab[k * n:].ctypes.data
        undo_slot = np.zeros(max(n, 1), dtype=np.int32)
        undo_val = np.zeros(max(n, 1), dtype=np.uint8)
        plan = _lib.DcpPlan()
        _lib.check(self.L.vlsfr_dcp_assign(self.lru._h, self.qp.ctypes.data, gl.ctypes.data, pl.ctypes.data, n,
                                           int(transactional), o(10), o(11), o(0), o(12), o(1), o(4), o(7),
                                           undo_slot.ctypes.data, undo_val.ctypes.data, ctypes.byref(plan)),
                   "vlsfr_dcp_assign")
        return tab, plan, (undo_slot, undo_val)

    def undo(self, plan, undo):
        _lib.check(self.L.vlsfr_dcp_undo(self.lru._h, self.qp.ctypes.data, undo[0].ctypes.data,
                                         undo[1].ctypes.data, ctypes.byref(plan)), "vlsfr_dcp_undo")

    # -------------------------------------------------------------------------------------------
    def run_pass(self, p, g, probe_label, gallery_label, transactional, row_offset=None):
        """p: [B, D] fp32 device tensor (may require grad); g: fp32 device tensor of gallery embeddings;
        labels: host int64 sequences / CPU tensors.

        Single process: g is [B, D] and the labels have B entries.  Data-parallel (row_offset given):
        g and the labels cover the whole batch of all ranks in rank order, p holds this rank's B
        rows starting at row_offset; every rank replays the same bookkeeping and pool writes, the
        returned loss is this rank's share of the global loss (sum over ranks = reference loss)."""
        if not p.is_cuda:
            raise _lib.VlsfrError("DcpHead.run_pass needs device tensors: the head has no CPU path")
        if torch.is_tensor(probe_label):
            probe_label = probe_label.cpu().numpy()
        if torch.is_tensor(gallery_label):
            gallery_label = gallery_label.cpu().numpy()
        B = int(p.shape[0])
        n = int(g.shape[0])
        r0 = 0 if row_offset is None else int(row_offset)
        assert p.shape == (B, self.D) and g.shape == (n, self.D) and r0 + B <= n
        tab, plan, undo = self.assign(probe_label, gallery_label, transactional)
        dev = p.device
        tab_d = torch.from_numpy(tab).pin_memory().to(dev, non_blocking=True)
        pd = p.detach().float().contiguous()
        gd = g.detach().float().contiguous()
        cfg = self._cfg(B, n if n != B else 0)
        ws = self._workspace(cfg, dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dP = torch.empty(B, self.D, dtype=torch.float32, device=dev)
        base = tab_d.data_ptr()
        at = lambda k, extra=0: ctypes.c_void_p(base + 4 * (k * n + extra))
        fn = self.L.vlsfr_head_fwd_bwd
        fn.restype = ctypes.c_int
        rc = fn(ctypes.byref(cfg), ctypes.c_void_p(pd.data_ptr()), ctypes.c_void_p(gd.data_ptr()),
                ctypes.c_void_p(self.queue.data_ptr()), at(0, r0), at(1), at(4), at(7), ctypes.c_int32(plan.n_special),
                ctypes.c_int32(plan.n_pos), ctypes.c_void_p(loss.data_ptr()), ctypes.c_void_p(dP.data_ptr()),
                ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), _stream_ptr())
        _lib.check(rc, "vlsfr_head_fwd_bwd")
        if transactional:
            self.undo(plan, undo)                                           # ffc.py:256-259
        else:
            sc = self.L.vlsfr_pool_scatter
            sc.restype = ctypes.c_int
            rc = sc(ctypes.c_void_p(self.queue.data_ptr()), ctypes.c_int64(self.Q), ctypes.c_int32(self.D),
                    ctypes.c_void_p(gd.data_ptr()), at(10), at(11), ctypes.c_int32(n), ctypes.c_int32(0),
                    ctypes.c_void_p(self.shadow.scatter_target()), _stream_ptr())   # ffc.py:182
            _lib.check(rc, "vlsfr_pool_scatter")
            self.shadow.after_scatter(at(11), n)
        self._keep = (tab_d, pd, gd)     # keep operands alive until the next pass is enqueued
        return _HeadFn.apply(p, loss.reshape(()), dP)


class ShardedDcpHead(object):
    """Identity-sharded Dynamic Class Pool: rank `rank` of `world` holds pool slots
    [rank * Qs, (rank + 1) * Qs) of the reference's queue[2, Q, D]; LRU / queue_position state is
    replicated (every rank replays the global label sequence).  One pass =
      partial()  — local sweep over the shard for ALL rows of the all-gathered batch
      combine()  — all-reduce(max) of the row maxima, all-gather of the hard-negative candidates,
                   rescale, all-reduce(sum) of (O, T, L, zt)          [3 small collectives]
      finish()   — loss and dL/dp rows from the combined state; the committing pass scatters g
                   into the owned slots.
    The collectives are passed in (`comm`), so the same code runs under torch.distributed (RCCL) and
    under the in-process simulator the single-GPU tests use."""

    def __init__(self, queue_shard, rank, world, Q_total, scale, margin, loss_type, precise=False, lru=None, qp=None):
        self.L = _lib.lib()
        _check_abi(self.L)
        self.queue = queue_shard                 # [2, Qs, D] fp32 device
        self.shadow = PoolShadow(queue_shard)
        self.rank, self.world = rank, world
        self.Qs, self.D, self.Q = int(queue_shard.shape[1]), int(queue_shard.shape[2]), int(Q_total)
        assert self.Qs * world == self.Q
        self.slot_lo = rank * self.Qs
        self.scale, self.margin, self.loss_type, self.precise = float(scale), float(margin), loss_type, bool(precise)
        self.head_dtype = os.environ.get('VLSFR_HEAD_DTYPE', 'bf16')
        self.hard_neg = min(max(int(self.Q * 0.0002), 3), 10)
        self.lru = lru if lru is not None else LRU(self.Q)
        self.qp = qp if qp is not None else np.zeros(self.Q, dtype=np.uint8)
        self._book = DcpHead.__new__(DcpHead)    # reuse the bookkeeping half
        self._book.L, self._book.lru, self._book.qp = self.L, self.lru, self.qp
        self._ws, self._ws_key = None, None
        self.n_chunks = int(os.environ.get('VLSFR_HEAD_CHUNKS', 0))     # column partition of the sweep (0 = by size)

    def _cfg(self, B):
        fp8 = self.head_dtype == "fp8"
        return HeadCfg(B, self.D, self.Qs, LOSS_TYPES[self.loss_type], self.scale, self.margin, self.hard_neg,
                       int(self.precise), self.n_chunks, self.slot_lo, 0, None if fp8 else self.shadow.ptr(not self.precise),
                       self.shadow.ptr(not self.precise, "fp8") if fp8 else None)

    def begin(self, p_all, g_all, probe_label, gallery_label, transactional):
        """Bookkeeping of one pass (identical on every rank) and, for SV, this rank's view of the hard-example
        thresholds st["thr"] [2, B], which the caller all-reduces (max) over the ranks before `sweep`."""
        B = int(p_all.shape[0])
        tab, plan, undo = self._book.assign(probe_label, gallery_label, transactional)
        dev = p_all.device
        tab_d = torch.from_numpy(tab).pin_memory().to(dev, non_blocking=True)
        cfg = self._cfg(B)
        key = (B, cfg.pool_bf16, cfg.pool_fp8, cfg.precise, cfg.loss_type, cfg.scale, cfg.n_chunks)
        if self._ws_key != key:
            fn = self.L.vlsfr_head_workspace_bytes
            fn.restype, fn.argtypes = ctypes.c_size_t, [ctypes.POINTER(HeadCfg)]
            self._ws = torch.empty(fn(ctypes.byref(cfg)), dtype=torch.uint8, device=dev)
            self._ws_key = key
        pd, gd = p_all.detach().float().contiguous(), g_all.detach().float().contiguous()
        st = dict(tab_d=tab_d, plan=plan, cfg=cfg, pd=pd, gd=gd, transactional=transactional, thr=None,
                  label=tab_d[:B])                      # pool_label block of the table (no second, blocking copy)
        if self.loss_type == "SV":
            base = tab_d.data_ptr()
            at = lambda k: ctypes.c_void_p(base + 4 * k * B)
            P = lambda t: ctypes.c_void_p(t.data_ptr())
            thr = torch.empty(2, B, dtype=torch.float32, device=dev)
            fn = self.L.vlsfr_head_shard_sv_thr
            fn.restype = ctypes.c_int
            _lib.check(fn(ctypes.byref(cfg), P(pd), P(gd), P(self.queue), at(0), at(1), at(4), at(7),
                          ctypes.c_int32(plan.n_special), P(thr), P(self._ws), ctypes.c_size_t(self._ws.numel()),
                          _stream_ptr()), "vlsfr_head_shard_sv_thr")
            st["thr"] = thr
        if transactional:
            self._book.undo(plan, undo)
        return st

    def _bufs(self, slot, B, dev):
        """Per-pass output buffers, kept across steps (two slots: the rollback and the committing pass of a step may be in
        flight together); nothing here is allocated per pass."""
        key = (B, str(dev), self.hard_neg)
        bufs = self.__dict__.setdefault("_pass_bufs", {})
        cur = bufs.get(slot)
        if cur is None or cur["key"] != key:
            f32 = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
            i32 = lambda *sh: torch.empty(*sh, dtype=torch.int32, device=dev)
            k = self.hard_neg
            cur = dict(key=key, packed=f32(B, 2, 2 * self.D + 2), M=f32(B, 2), cand_val=f32(B, 2, 10), cand_col=i32(B, 2, 10),
                       sel_col=i32(B, 2, k), sel_w=f32(B, 2, k), sel_loss=f32(B, 2), row_loss=f32(B, 2), loss=f32(1),
                       dP=f32(B, self.D))
            bufs[slot] = cur
        return cur

    def sweep(self, st):
        """Local sweep over this rank's slots for all rows (st["thr"] must hold the GLOBAL thresholds for SV); the per-row
        state lands in st["packed"] [B, 2, 2 D + 2] = (O | T | L | zt), the layout the ranks sum (vlsfr_head_shard_partial_packed)."""
        cfg, plan, tab_d, pd, gd = st["cfg"], st["plan"], st["tab_d"], st["pd"], st["gd"]
        B, dev = int(pd.shape[0]), pd.device
        bufs = self._bufs(0 if st["transactional"] else 1, B, dev)
        st.update(packed=bufs["packed"], M=bufs["M"], cand_val=bufs["cand_val"], cand_col=bufs["cand_col"], bufs=bufs)
        base, n = tab_d.data_ptr(), B
        at = lambda k: ctypes.c_void_p(base + 4 * k * n)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        thr = None
        if self.loss_type == "SV":
            thr = st["thr"] = st["thr"].contiguous()
        fixed = ctypes.c_int32(0)
        fn = self.L.vlsfr_head_shard_partial_packed
        fn.restype = ctypes.c_int
        _lib.check(fn(ctypes.byref(cfg), P(pd), P(gd), P(self.queue), at(0), at(1), at(4), at(7), ctypes.c_int32(plan.n_special),
                      ctypes.c_int32(plan.n_pos), P(thr) if thr is not None else None, P(st["packed"]), P(st["M"]),
                      P(st["cand_val"]), P(st["cand_col"]), ctypes.byref(fixed), P(self._ws), ctypes.c_size_t(self._ws.numel()),
                      _stream_ptr()), "vlsfr_head_shard_partial_packed")
        st["fixed_ref"] = bool(fixed.value)
        return st

    # views of the packed state (tests, tools)
    @staticmethod
    def unpack(st):
        D = (st["packed"].shape[2] - 2) // 2
        pk = st["packed"]
        return dict(O=pk[:, :, :D], T=pk[:, :, D:2 * D], L=pk[:, :, 2 * D], zt=pk[:, :, 2 * D + 1])

    def partial(self, p_all, g_all, probe_label, gallery_label, transactional, comm=None):
        st = self.begin(p_all, g_all, probe_label, gallery_label, transactional)
        if st["thr"] is not None:
            if comm is None:
                raise _lib.VlsfrError("ShardedDcpHead.partial: SV needs `comm` for the threshold all-reduce")
            st["thr"] = comm.all_reduce_max(st["thr"])
        return self.sweep(st)

    def combine(self, st, comm, own_rows=None):
        """The collectives of a pass.  comm: all_reduce_max(t), all_gather(t) -> [world, ...], and all_reduce_sum(t) or — with
        own_rows = (r0, r1), the rows of the gathered batch this rank's probe images produced — reduce_scatter_rows(t): the
        packed softmax state is then summed straight into its owner, and finish() yields the loss share and dL/dp of those
        rows only (half the bytes of the all-reduce, 1/W of the finish work).
        all-reduce(max): only when the fp32-pool sweep ran (st["fixed_ref"] False) — the shadow sweeps emit the state
        relative to the row's fixed reference exponent, which every rank computes identically from the probe row.
        all-gather of the hard-negative candidates: only when the batch has outlier rows."""
        B, k = int(st["M"].shape[0]), self.hard_neg
        plan, bufs, packed = st["plan"], st["bufs"], st["packed"]
        n_out = B - plan.n_pos
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        Mg = st["M"]
        if not st["fixed_ref"]:
            Mg = comm.all_reduce_max(st["M"].clone())
            w = torch.exp2(st["M"] - Mg)                                     # [B, 2] rescale to the global maximum
            w = torch.where(st["M"] <= -1e29, torch.zeros_like(w), w)
            D = self.D
            packed[:, :, :D] *= w.unsqueeze(2)
            packed[:, :, 2 * D] *= w
        sel_loss = None
        if n_out > 0:                                                        # hard negatives: global top-k of the candidates
            cv = comm.all_gather(st["cand_val"]).contiguous()                # [W, B, 2, 10]
            cc = comm.all_gather(st["cand_col"]).contiguous()
            fn = self.L.vlsfr_head_shard_topk_merge
            fn.restype = ctypes.c_int
            _lib.check(fn(ctypes.byref(st["cfg"]), P(cv), P(cc), ctypes.c_int32(int(cv.shape[0])), P(st["label"]),
                          ctypes.c_int32(n_out), P(bufs["sel_col"]), P(bufs["sel_w"]), P(bufs["sel_loss"]), _stream_ptr()),
                       "vlsfr_head_shard_topk_merge")
            base, n = st["tab_d"].data_ptr(), B
            at = lambda j: ctypes.c_void_p(base + 4 * j * n)
            fn = self.L.vlsfr_head_outlier_accum_strided
            fn.restype = ctypes.c_int
            _lib.check(fn(ctypes.byref(st["cfg"]), P(st["gd"]), P(self.queue), at(1), at(4), at(7), ctypes.c_int32(plan.n_special),
                          P(bufs["sel_col"]), P(bufs["sel_w"]), ctypes.c_int32(k),
                          ctypes.c_void_p(packed.data_ptr() + 4 * self.D), ctypes.c_int32(2 * self.D + 2), _stream_ptr()),
                       "vlsfr_head_outlier_accum_strided")
            sel_loss = bufs["sel_loss"]
            st["_keep"] = (cv, cc)
        if own_rows is None:
            summed = comm.all_reduce_sum(packed)
            own_rows = (0, B)
        else:
            summed = comm.reduce_scatter_rows(packed)
        st.update(Mg=Mg, summed=summed, sel_loss=sel_loss, rows=own_rows)
        return st

    def finish(self, st):
        """Loss and dL/dp of the rows combine() left on this rank: all rows of the batch (all-reduce form: the loss is
        then identical on every rank), or this rank's own rows (reduce-scatter form: the losses of the ranks sum to
        the reference loss) — one kernel (vlsfr_head_shard_finish) + the deterministic loss sum."""
        D, plan, bufs = self.D, st["plan"], st["bufs"]
        r0, r1 = st.get("rows", (0, int(st["M"].shape[0])))
        n = r1 - r0
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        summed = st["summed"]
        if summed.shape[0] != n:                                             # all-reduce form: every row is here
            summed = summed[r0:r1]
        Mg = st["Mg"][r0:r1]
        sel = st["sel_loss"][r0:r1] if st["sel_loss"] is not None else None
        dP, loss = bufs["dP"][:n], bufs["loss"]
        fn = self.L.vlsfr_head_shard_finish
        fn.restype = ctypes.c_int
        _lib.check(fn(ctypes.byref(st["cfg"]), P(summed), P(Mg), ctypes.c_void_p(st["label"].data_ptr() + 4 * r0),
                      P(sel) if sel is not None else None, ctypes.c_int32(n), ctypes.c_int32(plan.n_pos), P(bufs["row_loss"]),
                      P(loss), P(dP), _stream_ptr()), "vlsfr_head_shard_finish")
        if not st["transactional"]:
            sc = self.L.vlsfr_pool_scatter
            sc.restype = ctypes.c_int
            base, nb = st["tab_d"].data_ptr(), int(st["M"].shape[0])
            at = lambda j: ctypes.c_void_p(base + 4 * j * nb)
            _lib.check(sc(ctypes.c_void_p(self.queue.data_ptr()), ctypes.c_int64(self.Qs), ctypes.c_int32(D),
                          ctypes.c_void_p(st["gd"].data_ptr()), at(10), at(11), ctypes.c_int32(nb),
                          ctypes.c_int32(self.slot_lo),
                          ctypes.c_void_p(self.shadow.scatter_target()), _stream_ptr()),
                       "vlsfr_pool_scatter")
            self.shadow.after_scatter(at(11), nb, self.slot_lo)
        self._keep = st
        # the autograd node (_HeadFn) clones the loss itself and SAVES dP: dP gets its own storage here, the pass buffers are
        # rewritten by the next pass of this kind
        return loss.reshape(()), dP.clone()
