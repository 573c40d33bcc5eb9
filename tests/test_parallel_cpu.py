"""world_size-2 gloo test of the multi-process plumbing (CPU): label exchange over the gloo side
group, replicated LRU / queue_position bookkeeping on every rank, row offsets, and the flat-buffer
gradient reduction.  The device kernels are exercised by the GPU tests; here every rank checks that
the host side of the data-parallel pass is identical to the single-process bookkeeping on the
rank-order concatenation of the batches."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import vlsfr_amd  # noqa: F401
        from vlsfr_amd.head import DcpHead
        from vlsfr_amd.parallel import DataParallelFFC

        class Stub(object):          # the parts of FFC that DataParallelFFC touches on the host side
            def parameters(self):
                return []

            def buffers(self):
                return []

            def named_buffers(self):
                return []

        dp = DataParallelFFC(Stub(), dist)
        Q, D, B = 64, 16, 6
        rng = np.random.default_rng(100 + rank)
        head = DcpHead.__new__(DcpHead)          # host half only: allocator + bookkeeping, no device pool
        from vlsfr_amd import _lib
        from vlsfr_amd.lru import LRU
        head.L, head.Q, head.D = _lib.lib(), Q, D
        head.lru, head.qp = LRU(Q), np.zeros(Q, dtype=np.uint8)
        log = []
        for step in range(4):
            pl_local = rng.integers(0, 40, size=B).astype(np.int64)
            gl_local = rng.integers(0, 40, size=B).astype(np.int64)
            pl, gl = dp._gather_labels(torch.from_numpy(pl_local)), dp._gather_labels(gl_local)
            assert pl.shape == (world * B,) and np.array_equal(pl[rank * B:(rank + 1) * B], pl_local)
            for trans in (True, False):
                tab, plan, undo = head.assign(pl, gl, trans)
                n = world * B
                log.append((tab[:n].tolist(), tab[10 * n:12 * n].tolist(), plan.n_special, plan.n_pos))
                if trans:
                    head.undo(plan, undo)
        # flat gradient reduction
        flat = torch.full((10,), float(rank + 1))
        dist.all_reduce(flat)
        out.put((rank, log, head.lru.state_dict(), head.qp.tolist(), flat.tolist()))
    finally:
        dist.destroy_process_group()


def test_two_rank_bookkeeping_is_replicated():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, log0, lru0, qp0, flat0), (_, log1, lru1, qp1, flat1) = res
    assert log0 == log1 and lru0 == lru1 and qp0 == qp1          # every rank replays the same global sequence
    assert flat0 == [3.0] * 10 and flat1 == [3.0] * 10
    # and that sequence is the single-process bookkeeping on the concatenated batch
    from oracle.ffc_ref import dcp_assign_ref
    from oracle.lru_ref import LRURef
    Q, B = 64, 6
    rngs = [np.random.default_rng(100 + r) for r in range(world)]
    lru, qp = LRURef(Q), [0] * Q
    k = 0
    for step in range(4):
        pls, gls = [], []
        for r in range(world):
            pls.append(rngs[r].integers(0, 40, size=B))
            gls.append(rngs[r].integers(0, 40, size=B))
        pl, gl = np.concatenate(pls).tolist(), np.concatenate(gls).tolist()
        for trans in (True, False):
            rows, cols, labels, ones, saved = dcp_assign_ref(lru, qp, gl, pl, trans)
            assert log0[k][0] == labels and log0[k][1] == rows + cols
            k += 1
            if trans:
                for s, v in saved.items():
                    qp[s] = v
                lru.rollback_steps(len(gl))
    assert lru0 == lru.state_dict() and qp0 == qp
