"""Dynamic Class Pool head: host bookkeeping (native LRU) + the fused gfx950 head kernels.

One ``DcpHead.run_pass`` is one ``FFC.forward_impl`` (committing, reference ffc.py:153-204) or one
``FFC.forward_impl_rollback`` (transactional, ffc.py:208-260) *after* the two backbone calls: it
takes the probe/gallery embeddings and the two label vectors and returns the pass's loss as an
autograd node whose backward is the dL/dp the kernel already produced (the pool is a buffer, so
nothing else needs a gradient).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .lru import LRU

LOSS_TYPES = {"AM": 0, "Arc": 1, "SV": 2}


class HeadCfg(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("D", ctypes.c_int32), ("Q", ctypes.c_int64),
                ("loss_type", ctypes.c_int32), ("scale", ctypes.c_float), ("margin", ctypes.c_float),
                ("hard_neg", ctypes.c_int32), ("precise", ctypes.c_int32), ("n_chunks", ctypes.c_int32),
                ("n_rows_total", ctypes.c_int32)]


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
        self.queue = queue                       # [2, Q, D] fp32, device, contiguous (shared with FFC.queue)
        self.Q, self.D = int(queue.shape[1]), int(queue.shape[2])
        self.scale, self.margin, self.loss_type = float(scale), float(margin), loss_type
        self.precise, self.n_chunks = bool(precise), int(n_chunks)
        self.hard_neg = min(max(int(self.Q * 0.0002), 3), 10)          # ffc.py:48
        self.lru = LRU(self.Q)                                        # ffc.py:40
        self.qp = np.zeros(self.Q, dtype=np.uint8)                    # ffc.py:41-43
        self._ws = None
        self._ws_key = None

    # -------------------------------------------------------------------------------------------
    def _cfg(self, B, B_total=0):
        return HeadCfg(B, self.D, self.Q, LOSS_TYPES[self.loss_type], self.scale, self.margin, self.hard_neg,
                       int(self.precise), self.n_chunks, B_total)

    def _workspace(self, cfg, device):
        key = (cfg.B, cfg.n_rows_total, str(device))
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
                    ctypes.c_void_p(gd.data_ptr()), at(10), at(11), ctypes.c_int32(n), _stream_ptr())   # ffc.py:182
            _lib.check(rc, "vlsfr_pool_scatter")
        self._keep = (tab_d, pd, gd)     # keep operands alive until the next pass is enqueued
        return _HeadFn.apply(p, loss.reshape(()), dP)
