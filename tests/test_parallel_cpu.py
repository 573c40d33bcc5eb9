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


# ------------------------------------------------------------------------------------------------------------
# Partitioned SGD (ZeRO-1) and the collectives of the sharded step over gloo, world 2 and 4
# ------------------------------------------------------------------------------------------------------------
ZERO_SHAPES = [(16, 8, 3, 3), (37,), (5, 7), (1,), (130,), (64, 4, 1, 1), (9,)]      # odd sizes: padding inside and between buckets
ZERO_BUCKET = [0, 0, 1, 1, 2, 2, 2]


def _zero_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import vlsfr_amd  # noqa: F401
        from oracle import ffc_ref
        from vlsfr_amd.optim.fused import PartitionedSGD
        from vlsfr_amd.parallel import Comm
        comm = Comm(dist)
        g0 = torch.Generator().manual_seed(0)
        params = []
        for sh in ZERO_SHAPES:
            t = torch.randn(sh, generator=g0)
            if len(sh) == 4:
                t = t.contiguous(memory_format=torch.channels_last)
            params.append(torch.nn.Parameter(t))
        index = {id(p): i for i, p in enumerate(params)}
        opt = PartitionedSGD(params, 0.1, momentum=0.9, weight_decay=1e-4, nesterov=True, comm=comm,
                             bucket_of=lambda p: ZERO_BUCKET[index[id(p)]], n_buckets=3)

        # the device kernel of the shard update is csrc/optim.hip (GPU tests pin it); here the SAME rule from the oracle
        # is applied to the rank's slices, so the test covers layout, padding, reduce-scatter and all-gather plumbing
        def update_shards(group):
            part = opt._part
            bufs = [m if opt.__dict__.get("_stepped") else None for m in part["mshard"]]
            new = ffc_ref.sgd_nesterov_step_ref(part["pshard"], part["gshard"], bufs, group["lr"], group["momentum"],
                                                group["weight_decay"], nesterov=group["nesterov"])
            for m, b in zip(part["mshard"], new):
                if b is not m:
                    m.copy_(b)
            opt.__dict__["_stepped"] = True
        opt._update_shards = update_shards
        # reference: replicated update on the summed gradients
        ref = [p.detach().clone().double() for p in params]
        bufs = [None] * len(ref)
        gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
        for step in range(3):
            opt.zero_grad()
            grads_all = [[torch.randn(p.shape, generator=gens[r]) for p in params] for r in range(world)]
            for p, g in zip(params, grads_all[rank]):
                assert p.grad.data_ptr() % 16 == 0
                p.grad.copy_(g)
            for b in range(opt.n_buckets):
                opt.reduce_bucket(b)
            opt.step()
            summed = [sum(grads_all[r][i] for r in range(world)).double() for i in range(len(params))]
            bufs = ffc_ref.sgd_nesterov_step_ref(ref, summed, bufs, 0.1, 0.9, 1e-4)
        err = max(float((p.detach().double() - r).abs().max()) for p, r in zip(params, ref))
        opt.consolidate_state()
        merr = max(float((opt.state[p]["momentum_buffer"].double() - b).abs().max()) for p, b in zip(params, bufs))
        strides_ok = all(p.stride() == torch.empty(sh).contiguous(memory_format=torch.channels_last).stride()
                         for p, sh in zip(params, ZERO_SHAPES) if len(sh) == 4)
        # collectives used by the sharded head
        t = torch.arange(world * 3 * 2, dtype=torch.float32).reshape(world * 3, 2) * (rank + 1)
        rs = comm.reduce_scatter_rows(t)
        want = torch.arange(world * 3 * 2, dtype=torch.float32).reshape(world * 3, 2)[rank * 3:(rank + 1) * 3] * sum(range(1, world + 1))
        ag = comm.all_gather(torch.full((2,), float(rank)))
        out.put((rank, err, merr, strides_ok, bool(torch.equal(rs, want)), ag.tolist(),
                 float(comm.all_reduce(torch.tensor([float(rank)]), "max"))))
    finally:
        dist.destroy_process_group()


def _labels_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import vlsfr_amd  # noqa: F401
        from vlsfr_amd.parallel import DataParallelFFC

        class Stub(object):
            def parameters(self):
                return []

            def named_buffers(self):
                return []
        dp = DataParallelFFC(Stub(), dist)
        xl, yl = dp.exchange_labels(np.arange(4) + 10 * rank, torch.arange(4) + 100 * rank)
        out.put((rank, xl.tolist(), yl.tolist()))
    finally:
        dist.destroy_process_group()


def _spawn(target, world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=180) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world", [2, 4, 8])
def test_partitioned_sgd_equals_replicated_update(world):
    """ZeRO-1 step over gloo: bucketed reduce-scatter of the gradients, update of each rank's parameter / momentum slice,
    all-gather of the slices — three steps from random parameters and per-rank gradients equal the oracle's
    torch.optim.SGD rule on the summed gradients (parameters and momenta), on every rank.  World 8 is the node size the
    metric is quoted on: buckets are padded to multiples of 4 x 8 elements, and a 1- or 9-element parameter leaves most
    ranks' slices of its bucket empty."""
    for rank, err, merr, strides_ok, rs_ok, ag, mx in _spawn(_zero_worker, world):
        assert err < 1e-5 and merr < 1e-5, (rank, err, merr)
        assert strides_ok and rs_ok
        assert ag == [[float(r)] * 2 for r in range(world)] and mx == float(world - 1)


@pytest.mark.parametrize("world", [2, 8])
def test_label_exchange_is_rank_ordered(world):
    res = _spawn(_labels_worker, world)
    want_x = [v + 10 * r for r in range(world) for v in range(4)]
    want_y = [v + 100 * r for r in range(world) for v in range(4)]
    for rank, xl, yl in res:
        assert xl == want_x and yl == want_y


def test_rehearsal_stand_ins_have_the_collectives_shapes():
    """parallel.RehearsalDist / RehearsalComm (bench.py --rehearse-world: rank 0 of W in one process): every stand-in returns
    what the real collective returns in shape and dtype, own data in rank 0's position, and the drawn labels of the other
    ranks follow main.py:53-60 (an id half shared between the two views)."""
    from vlsfr_amd.parallel import RehearsalComm, RehearsalDist
    W, B = 8, 6
    d = RehearsalDist(W, 1000, seed=1)
    c = RehearsalComm(W)
    assert d.get_world_size() == W and d.get_rank() == 0 and c.world == W and c.rank == 0
    lab = torch.stack([torch.arange(B), torch.arange(B) + 50])
    out = [torch.empty_like(lab) for _ in range(W)]
    d.all_gather(out, lab)
    assert torch.equal(out[0], lab)
    for r in range(1, W):
        assert out[r].shape == (2, B) and torch.equal(out[r][0, :B // 2], out[r][1, :B // 2]) and int(out[r].max()) < 1000
        assert len(set(out[r][0, :B // 2].tolist())) == B // 2
    t = torch.randn(B, 5)
    g = c.all_gather(t)
    assert g.shape == (W, B, 5) and torch.equal(g[0], t) and torch.equal(g[3], t.roll(3, 0))
    assert torch.equal(c.reduce_scatter_rows(torch.arange(W * 2 * 3.0).reshape(W * 2, 3)), torch.arange(6.0).reshape(2, 3))
    flat, shard = torch.zeros(W * 4), torch.ones(4)
    c.all_gather_into(flat, shard)
    assert flat[:4].tolist() == [1.0] * 4 and float(flat[4:].abs().sum()) == 0.0
    o = torch.empty(4)
    c.reduce_scatter_sum(o, torch.arange(32.0))
    assert o.tolist() == [0.0, 1.0, 2.0, 3.0]
    assert c.all_reduce_max(t) is t and c.all_reduce_sum(t) is t and c.broadcast(t) is t
