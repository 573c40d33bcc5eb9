"""Multi-GPU FFC step: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference has no distributed code; the semantics defined here are
*result == single-process reference on the rank-order concatenation of all ranks' batches*
(BatchNorm statistics stay per rank, like DistributedDataParallel without SyncBN):

  * backbones: data parallel — every rank runs its own B rows; parameter gradients are summed
    with one all-reduce over a flat gradient buffer (the loss normalisers are already global, so the
    sum is the reference gradient);
  * Dynamic Class Pool, two forms:
      - `ShardedFFC` (default for N > 1): the pool is split by slot range over the ranks
        (queue[2, Q/W, D] per GPU).  p, g are all-gathered over RCCL, the labels over a gloo side group
        (no device sync); every rank replays the identical LRU bookkeeping, sweeps ITS slots for ALL
        rows, and the softmax state is combined with all-reduce(max) + all-reduce(sum) of
        (O, T, L, zt) (head.ShardedDcpHead) — three small collectives per pass.
      - `DataParallelFFC`: replicated pool, every rank sweeps the whole pool for its own rows; no
        softmax collective (the A/B baseline, and the fallback when the pool does not divide over the ranks).
        SV under the sharded pool adds one small all-reduce(max) of the hard-example thresholds per pass.
"""
import numpy as np
import torch


class DataParallelFFC(object):
    def __init__(self, model, dist):
        self.m = model
        self.dist = dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.cpu_group = dist.new_group(backend="gloo")
        self.rccl = dist.get_backend() == "nccl"   # "nccl" IS RCCL on ROCm; gloo only in the 1-GPU rehearsal
        self._flat = None
        # identical starting point on every rank
        pre_sharded = getattr(model, 'pool_shard', None) is not None   # every rank built its own slots (ffc.build_pool)
        for t in list(model.parameters()) + [b for n, b in model.named_buffers() if not (pre_sharded and n == 'queue')]:
            if self.rccl:
                dist.broadcast(t.data, src=0)
            else:
                c = t.data.cpu()
                dist.broadcast(c, src=0)
                t.data.copy_(c)

    def _gather_labels(self, lab):
        lab = torch.as_tensor(lab, dtype=torch.int64).cpu().contiguous()
        out = [torch.empty_like(lab) for _ in range(self.world)]
        self.dist.all_gather(out, lab, group=self.cpu_group)
        return torch.cat(out).numpy()

    def _gather_rows(self, g):
        if self.rccl:
            out = torch.empty(self.world * g.shape[0], g.shape[1], dtype=g.dtype, device=g.device)
            self.dist.all_gather_into_tensor(out, g.contiguous())
            return out
        parts = [torch.empty(g.shape, dtype=g.dtype) for _ in range(self.world)]   # rehearsal path (gloo, host)
        self.dist.all_gather(parts, g.cpu().contiguous())
        return torch.cat(parts).to(g.device)

    def _pass(self, p_data, g_data, probe_label, gallery_label, transactional):
        m = self.m
        head = m._ensure_head()
        p, g = m.embed_pair(p_data, g_data, update_gallery=transactional)
        with torch.no_grad():
            g_all = self._gather_rows(g)
        pl = self._gather_labels(probe_label)
        gl = self._gather_labels(gallery_label)
        return head.run_pass(p, g_all, pl, gl, transactional, row_offset=self.rank * p.shape[0])

    def __call__(self, x, y, x_label, y_label):
        loss2 = self._pass(x, y, x_label, y_label, True)      # ffc.py:265
        loss1 = self._pass(y, x, y_label, x_label, False)     # ffc.py:266
        return loss1 + loss2

    def reduce_gradients(self, optimizer=None):
        """Sum the probe-net gradients over ranks (one all-reduce on a flat buffer)."""
        flat = optimizer.flat_grad() if optimizer is not None and hasattr(optimizer, "flat_grad") else None
        if flat is not None:
            if self.rccl:
                self.dist.all_reduce(flat)
            else:
                c = flat.cpu()
                self.dist.all_reduce(c)
                flat.copy_(c)
            return
        grads = [p.grad for p in self.m.probe_net.parameters() if p.requires_grad and p.grad is not None]
        flat = torch._utils._flatten_dense_tensors(grads)
        self.dist.all_reduce(flat)
        for g, s in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
            g.copy_(s)

    def global_loss(self, loss):
        t = loss.detach().clone()
        self.dist.all_reduce(t)
        return t


class _DistComm(object):
    """The three collectives of head.ShardedDcpHead over torch.distributed (RCCL; gloo rehearsal via host)."""

    def __init__(self, dist, rccl):
        self.dist, self.rccl, self.world = dist, rccl, dist.get_world_size()

    def _reduce(self, t, op):
        if self.rccl:
            self.dist.all_reduce(t, op=op)
            return t
        c = t.cpu()
        self.dist.all_reduce(c, op=op)
        return c.to(t.device)

    def all_reduce_max(self, t):
        return self._reduce(t, self.dist.ReduceOp.MAX)

    def all_reduce_sum(self, t):
        return self._reduce(t, self.dist.ReduceOp.SUM)

    def all_gather(self, t):
        t = t.contiguous()
        if self.rccl:
            out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            self.dist.all_gather_into_tensor(out, t)
            return out
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(self.world)]
        self.dist.all_gather(parts, t.cpu())
        return torch.stack(parts).to(t.device)


class ShardedFFC(DataParallelFFC):
    """Identity-sharded pool: after construction `model.queue` holds only this rank's slots
    [rank * Q / W, (rank + 1) * Q / W) (use `gather_pool()` for a checkpoint)."""

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
        self.comm = _DistComm(dist, self.rccl)

    def gather_pool(self):
        return self.comm.all_gather(self.head.queue).permute(1, 0, 2, 3).reshape(2, -1, self.head.D)

    def _pass(self, p_data, g_data, probe_label, gallery_label, transactional):
        from .head import _HeadFn
        m = self.m
        p, g = m.embed_pair(p_data, g_data, update_gallery=transactional)
        with torch.no_grad():
            g_all = self._gather_rows(g)
            p_all = self._gather_rows(p.detach())
        pl = self._gather_labels(probe_label)
        gl = self._gather_labels(gallery_label)
        st = self.head.partial(p_all, g_all, pl, gl, transactional, comm=self.comm)
        st = self.head.combine(st, self.comm)
        loss, dP = self.head.finish(st)
        B = p.shape[0]
        # every rank holds the same global loss; its autograd edge carries this rank's rows of dL/dp
        return _HeadFn.apply(p, loss.reshape(()), dP[self.rank * B:(self.rank + 1) * B].contiguous())
