"""Multi-GPU FFC step: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference has no distributed code; the semantics defined here are
*result == single-process reference on the rank-order concatenation of all ranks' batches*
(BatchNorm statistics stay per rank, like DistributedDataParallel without SyncBN):

  * backbones: data parallel — every rank runs its own B rows.  Parameter gradients are summed bucket by bucket
    with reduce-scatter on a communication stream WHILE the backward pass still runs (the backbone executor
    records one event per bucket, csrc/iresnet.cpp vlsfr_iresnet_backward_staged); each rank then updates its
    1/W slice of the parameters against its slice of the momentum (optim.fused.PartitionedSGD, ZeRO-1) and the
    updated slices are all-gathered.  The loss normalisers are global, so the summed gradient is the reference's.
  * Dynamic Class Pool, two forms:
      - `ShardedFFC` (default for N > 1): the pool is split by slot range over the ranks (queue[2, Q/W, D] per
        GPU, built shard-local by ffc.build_pool).  Per pass: ONE all-gather of the packed (p | g) rows; every rank
        replays the identical LRU bookkeeping and sweeps ITS slots for ALL W*B rows (head.ShardedDcpHead); the
        per-row softmax states are combined with an all-reduce(max) of the reference exponents [W*B, 2] and ONE
        reduce-scatter(sum) of the packed (O, T, L, zt) rows, after which every rank holds the loss terms and dL/dp
        of ITS OWN rows only; an all-gather of top-k candidates is added only when a batch has outlier rows, and SV
        adds one small all-reduce(max) of the hard-example thresholds.
      - `DataParallelFFC`: replicated pool, every rank sweeps the whole pool for its own rows; no softmax
        collective (the A/B baseline, and the fallback when the pool does not divide over the ranks).
  * labels are host arrays in the reference (main.py:53-60) and the LRU bookkeeping is host work, so they are
    exchanged once per step over a gloo side group: a device collective would force the host to wait for the
    stream before it can run the bookkeeping, i.e. serialise kernel issue with kernel execution.

The N > 1 path has been exercised with 2 ranks over gloo (CPU tests; one-GPU rehearsal in tests/test_parallel_gpu.py);
RCCL itself runs in the driver's multi-GPU bench only (no multi-GPU box is available to the build).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib


def warm_stream_pool(device, n=4):
    """Call BEFORE torch.distributed.init_process_group(..., device_id=...): creates torch's stream pool (and runs one tiny
    kernel on a few of its streams) so that the hardware queues HIP deals out in creation order go to the streams the step
    uses side by side, not to the communicator's.  Measured with one rank over RCCL, ir100 / 10 M identities / batch 256:
    92.6 ms per step with the pool first, 101 ms with RCCL's streams first."""
    if not torch.cuda.is_available():
        return
    streams = [torch.cuda.Stream(device=device) for _ in range(n)]
    for st in streams:
        with torch.cuda.stream(st):
            torch.zeros(1, device=device)
    torch.cuda.synchronize(device)


_masked_streams = []   # (keeps the HIP stream handles of cu_masked_stream alive)


def cu_masked_stream(device, reserve, low):
    """A HIP stream whose kernels may run on all CUs but `reserve` of them (hipExtStreamCreateWithCUMask), as a torch stream.
    Mask bit b is CU b / 8 of XCD b % 8 on the 8-XCD parts, so clearing the lowest (low=True) or the highest `reserve` bits takes
    reserve / 8 CUs out of every XCD.  Two chains that run side by side get complementary masks: each has CUs the other chain's
    convolutions never occupy, for its BatchNorm kernels (experiment: bench.py --cu-reserve)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    props_cus = torch.cuda.get_device_properties(device).multi_processor_count
    words = (props_cus + 31) // 32
    bits = [1] * props_cus
    rng = range(0, reserve) if low else range(props_cus - reserve, props_cus)
    for b in rng:
        bits[b] = 0
    mask = (ctypes.c_uint32 * words)()
    for b, v in enumerate(bits):
        if v:
            mask[b // 32] |= (1 << (b % 32))
    st = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    ext = torch.cuda.ExternalStream(st.value, device=device)
    _masked_streams.append((st, ext))
    return ext


class Comm(object):
    """The collectives of the step on device tensors.  Backend "nccl" (= RCCL on ROCm) runs them in place; under
    gloo (CPU tests, the 2-rank rehearsal on a 1-GPU box) the same calls stage through host memory."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.rccl = dist.get_backend(group) == "nccl"
        # diagnostic (VLSFR_COMM_SOLO=1, one rank): collectives become local copies — separates what the collectives cost
        # from what the rest of the distributed step costs
        self.solo = self.world == 1 and os.environ.get("VLSFR_COMM_SOLO", "0") == "1"

    def _host(self, t):
        return t if not t.is_cuda or self.rccl else t.cpu()

    def broadcast(self, t, src=0):
        if self.solo:
            return t
        c = self._host(t)
        self.dist.broadcast(c, src=src, group=self.group)
        if c is not t:
            t.copy_(c)
        return t

    def all_reduce(self, t, op="sum"):
        rop = self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM
        if self.solo:
            return t
        c = self._host(t)
        self.dist.all_reduce(c, op=rop, group=self.group)
        if c is not t:
            t.copy_(c)
        return t

    def all_gather(self, t):
        """-> [world, *t.shape]"""
        t = t.contiguous()
        if self.solo:
            return t.unsqueeze(0).clone()
        if self.rccl or not t.is_cuda:
            out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            if self.rccl:
                self.dist.all_gather_into_tensor(out, t, group=self.group)
            else:
                self.dist.all_gather(list(out.unbind(0)), t, group=self.group)
            return out
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(self.world)]
        self.dist.all_gather(parts, t.cpu(), group=self.group)
        return torch.stack(parts).to(t.device)

    def all_gather_into(self, out, shard):
        """out (flat, world * shard.numel()) <- concatenation of every rank's shard."""
        if self.solo:
            out.copy_(shard.reshape(-1))
            return out
        if self.rccl:
            self.dist.all_gather_into_tensor(out, shard, group=self.group)
            return out
        out.copy_(self.all_gather(shard).reshape(-1))
        return out

    def reduce_scatter_sum(self, out, inp):
        """out <- this rank's 1/world slice (along dim 0) of the sum of `inp` over the ranks."""
        if self.solo:
            out.copy_(inp)
            return out
        if self.rccl:
            self.dist.reduce_scatter_tensor(out, inp, group=self.group)
            return out
        c = inp.cpu() if inp.is_cuda else inp.clone()          # gloo has no reduce-scatter: all-reduce and slice
        self.dist.all_reduce(c, group=self.group)
        n = c.shape[0] // self.world
        out.copy_(c[self.rank * n:(self.rank + 1) * n])
        return out

    # the names head.ShardedDcpHead uses
    def all_reduce_max(self, t):
        return self.all_reduce(t, "max")

    def all_reduce_sum(self, t):
        return self.all_reduce(t, "sum")

    def reduce_scatter_rows(self, t):
        out = torch.empty((t.shape[0] // self.world,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        return self.reduce_scatter_sum(out, t.contiguous())


class RehearsalDist(object):
    """Stand-in for the torch.distributed module that lets ONE process run rank 0's step of a `world`-rank job
    (bench.py --rehearse-world, tests): shard-local pool, gathered batch of world * B rows, replicated LRU replaying the
    global label sequence, partitioned optimizer — with every collective replaced by a local operation of the same
    shape (RehearsalComm), so what is measured is rank 0's compute at the node's shapes WITHOUT wire time.  The other
    ranks' labels are drawn here (main.py:53-60 structure: an id half shared between the two views, an instance half)."""

    class ReduceOp(object):
        MAX, SUM = "max", "sum"

    def __init__(self, world, n_id, seed=0):
        self.world, self.n_id = int(world), int(n_id)
        self.rng = np.random.default_rng(seed)

    def get_world_size(self, group=None):
        return self.world

    def get_rank(self, group=None):
        return 0

    def get_backend(self, group=None):
        return "rehearsal"

    def new_group(self, backend=None):
        return None

    def barrier(self, group=None):
        pass

    def broadcast(self, t, src=0, group=None):
        return t

    def all_reduce(self, t, op=None, group=None):
        return t

    def all_gather(self, out, t, group=None):
        """Only the host label exchange comes here (DataParallelFFC.exchange_labels: t = [2, B] int64)."""
        out[0].copy_(t)
        B = t.shape[-1]
        h = B // 2
        for r in range(1, self.world):
            ids = self.rng.choice(self.n_id, size=h, replace=False)
            for v in range(2):
                out[r][v] = torch.from_numpy(np.concatenate([ids, self.rng.integers(0, self.n_id, size=B - h)]).astype(np.int64))


class RehearsalComm(object):
    """Comm with the wire removed (see RehearsalDist): gathers tile this rank's tensor (rows rolled, so the other ranks'
    embedding rows are different unit vectors), reductions keep this rank's contribution, scatters keep its slice."""

    def __init__(self, world):
        self.world, self.rank, self.rccl, self.solo = int(world), 0, False, False

    def broadcast(self, t, src=0):
        return t

    def all_reduce(self, t, op="sum"):
        return t

    all_reduce_max = all_reduce_sum = lambda self, t: t

    def all_gather(self, t):
        return torch.stack([t.roll(r, 0) if t.dim() > 0 and t.shape[0] > 1 else t for r in range(self.world)])

    def all_gather_into(self, out, shard):
        out[:shard.numel()].copy_(shard.reshape(-1))          # the other ranks' slices keep their (valid, older) contents
        return out

    def reduce_scatter_sum(self, out, inp):
        out.copy_(inp[:out.shape[0]])
        return out

    def reduce_scatter_rows(self, t):
        return t[:t.shape[0] // self.world].contiguous()


class DataParallelFFC(object):
    """Data-parallel backbones over a replicated pool; also the base of ShardedFFC (everything but the head)."""
    overlap_head = os.environ.get("VLSFR_OVERLAP_HEAD", "0") == "1"

    def __init__(self, model, dist):
        self.m = model
        self.dist = dist
        self.comm = RehearsalComm(dist.world) if isinstance(dist, RehearsalDist) else Comm(dist)
        self.world, self.rank = self.comm.world, self.comm.rank
        self.rccl = self.comm.rccl
        self.cpu_group = dist.new_group(backend="gloo")        # labels (host arrays) travel here, see module docstring
        self._labels = None
        self._comm_stream = None
        # The step's own side streams are created BEFORE the first collective creates RCCL's: HIP deals streams to its
        # (few) hardware queues in creation order, and two streams on one queue run one after the other — the gallery
        # pass, the head sweep and the second backward pass must not end up behind a collective's stream.
        if torch.cuda.is_available() and next(model.parameters()).is_cuda:
            d = next(model.parameters()).device
            if model.__dict__.get('_side_stream') is None:
                model.__dict__['_side_stream'] = torch.cuda.Stream(device=d)
            self._head_stream = torch.cuda.Stream(device=d)
            pn = getattr(model, "probe_net", None)
            if pn is not None and hasattr(pn, "_side_stream") and callable(pn._side_stream):
                pn._side_stream(d)
            self._comm_stream = torch.cuda.Stream(device=d)
        # identical starting point on every rank (a pool built shard-local is already consistent by construction)
        pre_sharded = getattr(model, 'pool_shard', None) is not None
        for t in list(model.parameters()) + [b for n, b in model.named_buffers() if not (pre_sharded and n == 'queue')]:
            self.comm.broadcast(t.data)
        head = getattr(model, "_head", None)
        if head is not None:                                   # the broadcast wrote the pool through .data (in place under RCCL):
            head.shadow.invalidate()                           # torch's version counter did not move, the sweep's mirror is stale
        for net in (getattr(model, "probe_net", None),):
            if net is not None:
                net.__dict__["signal_stages"] = True           # the backward passes record their bucket events

    # ---- the optimizer of the step ----------------------------------------------------------------
    def make_optimizer(self, lr, momentum=0.9, weight_decay=1e-4, nesterov=True):
        """The partitioned SGD over the probe net's trainable parameters, bucketed as the backbone's backward pass
        completes them."""
        from .optim.fused import PartitionedSGD
        net = self.m.probe_net
        names = {id(p): n for n, p in net.named_parameters()}
        params = [p for p in self.m.parameters() if p.requires_grad]
        return PartitionedSGD(params, lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov, comm=self.comm,
                              bucket_of=lambda p: net.bucket_of(names[id(p)]), n_buckets=net.N_BUCKETS)

    def _stream(self, device):
        if self._comm_stream is None or self._comm_stream.device != device:
            self._comm_stream = torch.cuda.Stream(device=device)
        return self._comm_stream

    def reduce_gradients(self, optimizer=None):
        """Call right after loss.backward() (which only ENQUEUES the backward passes): per bucket, the communication
        stream waits for the events both backward passes record when the bucket's gradients are complete, then
        reduce-scatters it — the reduction of layer4 overlaps the backward pass of layer3, and so on.  The main
        stream waits for the communication stream before optimizer.step()."""
        from .optim.fused import PartitionedSGD
        if not isinstance(optimizer, PartitionedSGD):
            return self._all_reduce_gradients(optimizer)
        net = self.m.probe_net
        main = torch.cuda.current_stream()
        comm_s = self._stream(main.device)
        L = _lib.lib()
        evs = [net.stage_events(slot) for slot in (0, 1)]
        for b in range(optimizer.n_buckets):
            for slot in (0, 1):
                _lib.check(L.vlsfr_stream_wait_event(ctypes.c_void_p(comm_s.cuda_stream), ctypes.c_void_p(evs[slot][b])),
                           "vlsfr_stream_wait_event")
            with torch.cuda.stream(comm_s):
                optimizer.reduce_bucket(b)
        main.wait_stream(comm_s)

    def _all_reduce_gradients(self, optimizer=None):
        """Replicated update (FusedSGD / any torch optimizer): one all-reduce over the flat gradient buffer."""
        flat = optimizer.flat_grad() if optimizer is not None and hasattr(optimizer, "flat_grad") else None
        if flat is not None:
            self.comm.all_reduce(flat)
            return
        grads = [p.grad for p in self.m.probe_net.parameters() if p.requires_grad and p.grad is not None]
        flat = torch._utils._flatten_dense_tensors(grads)
        self.comm.all_reduce(flat)
        for g, s in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
            g.copy_(s)

    # ---- one step -----------------------------------------------------------------------------------
    def exchange_labels(self, x_label, y_label):
        """Both label vectors of every rank in one gloo all-gather -> (x_labels, y_labels) of the whole batch."""
        lab = torch.stack([torch.as_tensor(x_label, dtype=torch.int64).cpu(), torch.as_tensor(y_label, dtype=torch.int64).cpu()])
        out = [torch.empty_like(lab) for _ in range(self.world)]
        self.dist.all_gather(out, lab.contiguous(), group=self.cpu_group)
        allv = torch.stack(out)                                   # [W, 2, B]
        return allv[:, 0].reshape(-1).numpy(), allv[:, 1].reshape(-1).numpy()

    def _gather_labels(self, lab):                                # one vector (tests, tools)
        lab = torch.as_tensor(lab, dtype=torch.int64).cpu().contiguous()
        out = [torch.empty_like(lab) for _ in range(self.world)]
        self.dist.all_gather(out, lab, group=self.cpu_group)
        return torch.cat(out).numpy()

    def _gather_rows(self, t):
        return self.comm.all_gather(t).reshape((-1,) + tuple(t.shape[1:]))

    def _head_pass(self, p, g, probe_label, gallery_label, transactional):
        """Everything of a pass after the two backbones (collectives included); runs on the current stream."""
        head = self.m._ensure_head()
        with torch.no_grad():
            g_all = self._gather_rows(g)
        return head.run_pass(p, g_all, probe_label, gallery_label, transactional, row_offset=self.rank * p.shape[0])

    def _mark(self, name):
        marks = self.__dict__.get('_marks')          # diagnostic: bench.py --phases (main-stream events)
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e))

    def _pass(self, p_data, g_data, probe_label, gallery_label, transactional):
        self._mark("start" if transactional else "head of pass 1")
        p, g = self.m.embed_pair(p_data, g_data, update_gallery=transactional)
        self._mark("backbones of pass %d" % (1 if transactional else 2))
        out = self._head_pass(p, g, probe_label, gallery_label, transactional)
        if not transactional:
            self._mark("head of pass 2")
        return out

    def __call__(self, x, y, x_label, y_label):
        xl, yl = self.exchange_labels(x_label, y_label)
        m = self.m
        if not self.overlap_head or not m.__dict__.get('concurrent_streams', True) or not x.is_cuda:
            loss2 = self._pass(x, y, xl, yl, True)        # ffc.py:265
            loss1 = self._pass(y, x, yl, xl, False)       # ffc.py:266
            return loss1 + loss2
        # overlap_head (VLSFR_OVERLAP_HEAD=1; off by default): as FFC.forward, the head of the rollback pass (gather, sweep,
        # combine: collectives included) on a head stream beside the backbones of the commit pass.  Both heads run on that
        # one stream in program order, so pool / LRU state and the order of the collectives are those of the sequential
        # schedule on every rank.  Measured with one rank over RCCL (bench.py --force-dist, ir100 / 10 M identities):
        # 92.5-95.0 ms per step against 92.1-92.3 with the heads on the main stream — sweep and backbones are both
        # MFMA-bound, running them side by side returns nothing.
        main = torch.cuda.current_stream()
        hs = self.__dict__.get('_head_stream')
        if hs is None or hs.device != main.device:
            hs = self._head_stream = torch.cuda.Stream(device=main.device)
        p1, g1 = m.embed_pair(x, y, update_gallery=True)
        hs.wait_stream(main)
        with torch.cuda.stream(hs):
            loss2 = self._head_pass(p1, g1, xl, yl, True)
        p1.record_stream(hs)
        g1.record_stream(hs)
        p2, g2 = m.embed_pair(y, x, update_gallery=False)
        hs.wait_stream(main)
        with torch.cuda.stream(hs):
            loss1 = self._head_pass(p2, g2, yl, xl, False)
            total = loss1 + loss2
        p2.record_stream(hs)
        g2.record_stream(hs)
        main.wait_stream(hs)
        total.record_stream(main)
        return total

    def global_loss(self, loss):
        """Every rank returns its share of the loss (the shares sum to the reference loss); this is the sum."""
        return self.comm.all_reduce(loss.detach().clone())


class ShardedFFC(DataParallelFFC):
    """Identity-sharded pool: `model.queue` holds only this rank's slots [rank * Q / W, (rank + 1) * Q / W)
    (`pool_state()` / `load_pool_state()` for shard-wise checkpoints, `gather_pool()` for a reference-format one)."""

    def __init__(self, model, dist):
        super(ShardedFFC, self).__init__(model, dist)
        from .head import ShardedDcpHead
        Q = model.queue_size
        if Q % self.world:
            raise ValueError("queue_size must be divisible by the number of ranks for the sharded pool")
        Qs = Q // self.world
        if getattr(model, 'pool_shard', None) is not None:
            if tuple(model.pool_shard) != (self.rank, self.world) or model.queue.shape[1] != Qs:
                raise ValueError("FFC was built with pool_shard=%r, this process is rank %d of %d" %
                                 (model.pool_shard, self.rank, self.world))
            shard = model.queue                  # built shard-local: no rank ever held the whole pool
        else:
            shard = model.queue[:, self.rank * Qs:(self.rank + 1) * Qs].contiguous()
        state = model._state()
        model.queue = shard                      # releases the full replica
        model._head = None
        self.head = ShardedDcpHead(shard, self.rank, self.world, Q, model.scale, model.margin, model.loss_type,
                                   precise=model.precise_head, lru=state.lru, qp=state.qp)
        if model.__dict__.get('head_dtype'):
            self.head.head_dtype = model.__dict__['head_dtype']

    def gather_pool(self):
        """The whole pool [2, Q, D] on every rank (a collective; reference checkpoint format, main.py:85)."""
        return self.comm.all_gather(self.head.queue).permute(1, 0, 2, 3).reshape(2, -1, self.head.D)

    def pool_state(self):
        """This rank's part of a shard-wise checkpoint: its pool slots and the (replicated) allocator state."""
        keys, slots = self.head.lru.state_arrays()
        return dict(rank=self.rank, world=self.world, slot_lo=self.head.slot_lo, fc_shard=self.head.queue.cpu(),
                    lru_keys=torch.from_numpy(keys.copy()), lru_slots=torch.from_numpy(slots.copy()),
                    qp=torch.from_numpy(self.head.qp.copy()))

    def load_pool_state(self, st):
        if (int(st["rank"]), int(st["world"])) != (self.rank, self.world):
            raise ValueError("pool shard of rank %d/%d loaded on rank %d/%d" % (st["rank"], st["world"], self.rank, self.world))
        with torch.no_grad():
            self.head.queue.copy_(st["fc_shard"].to(self.head.queue.device))
        self.head.lru.reset()
        self.head.lru.restore_arrays(st["lru_keys"].numpy(), st["lru_slots"].numpy())
        self.head.qp[:] = st["qp"].numpy()

    overlap_head = os.environ.get("VLSFR_OVERLAP_HEAD", "0") == "1"

    def __call__(self, x, y, x_label, y_label):
        """overlap_head = True: the step with the SWEEP of the rollback pass (the one multi-millisecond kernel of the head,
        no collective in it) on a head stream beside the backbones of the commit pass; every collective stays on the main
        stream in the sequential schedule's order.  Off by default: with one rank over RCCL (bench.py --force-dist) 92.5-93.0 ms
        per step against 92.3 ms sequential (the sweep lengthens the MFMA-bound backbones beside it by its own duration)."""
        m = self.m
        if not self.overlap_head or not m.__dict__.get('concurrent_streams', True) or not x.is_cuda:
            return super(ShardedFFC, self).__call__(x, y, x_label, y_label)
        from .head import _HeadFn
        xl, yl = self.exchange_labels(x_label, y_label)
        head, comm = self.head, self.comm
        main = torch.cuda.current_stream()
        hs = self.__dict__.get('_head_stream')
        if hs is None or hs.device != main.device:
            hs = self._head_stream = torch.cuda.Stream(device=main.device)

        def gather(p, g):
            D = p.shape[1]
            with torch.no_grad():
                pg = self._gather_rows(torch.cat([p.detach(), g], dim=1))
                return pg[:, :D].contiguous(), pg[:, D:].contiguous()

        def begin(p_all, g_all, pl, gl, transactional):
            st = head.begin(p_all, g_all, pl, gl, transactional)
            if st["thr"] is not None:
                st["thr"] = comm.all_reduce_max(st["thr"])
            return st

        def tail(st, p):
            B = p.shape[0]
            st = head.combine(st, comm, own_rows=(self.rank * B, (self.rank + 1) * B))
            loss, dP = head.finish(st)
            return _HeadFn.apply(p, loss.reshape(()), dP.contiguous())

        marks = self.__dict__.get('_marks')                             # diagnostic: bench.py --phases (main-stream events)

        def mark(name):
            if marks is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append((name, e))

        mark("start")
        p1, g1 = m.embed_pair(x, y, update_gallery=True)
        mark("backbones of pass 1")
        st1 = begin(*gather(p1, g1), xl, yl, True)                       # ffc.py:265 (bookkeeping incl. its rollback)
        mark("gather + bookkeeping 1")
        before = set(id(v) for v in st1.values() if torch.is_tensor(v))
        hs.wait_stream(main)
        with torch.cuda.stream(hs):
            st1 = head.sweep(st1)
        for v in st1.values():
            if torch.is_tensor(v):
                v.record_stream(main if id(v) not in before else hs)
        p2, g2 = m.embed_pair(y, x, update_gallery=False)
        mark("backbones of pass 2 (sweep 1 beside them)")
        pa2, ga2 = gather(p2, g2)
        main.wait_stream(hs)
        mark("gather 2 + join of sweep 1")
        loss2 = tail(st1, p1)
        mark("combine + finish 1")
        st2 = head.sweep(begin(pa2, ga2, yl, xl, False))                 # ffc.py:266
        mark("bookkeeping + sweep 2")
        loss1 = tail(st2, p2)
        mark("combine + finish 2")
        return loss1 + loss2

    def _head_pass(self, p, g, probe_label, gallery_label, transactional):
        from .head import _HeadFn
        B, D = p.shape
        with torch.no_grad():
            pg = self._gather_rows(torch.cat([p.detach(), g], dim=1))        # ONE all-gather: [W * B, 2 D]
            p_all, g_all = pg[:, :D].contiguous(), pg[:, D:].contiguous()
        st = self.head.partial(p_all, g_all, probe_label, gallery_label, transactional, comm=self.comm)
        st = self.head.combine(st, self.comm, own_rows=(self.rank * B, (self.rank + 1) * B))
        loss, dP = self.head.finish(st)
        # this rank's share of the loss (its own rows); the autograd edge carries its rows of dL/dp
        return _HeadFn.apply(p, loss.reshape(()), dP.contiguous())
