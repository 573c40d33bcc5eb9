"""FFC module — same constructor, attributes and call contract as the reference ``ffc.FFC``
(ffc.py:10-267): probe net (trained) + gallery net (EMA copy, frozen), Dynamic Class Pool
``queue[2, Q, D]``, LRU slot allocator, ``queue_position_dict``, AM / Arc / SV margin losses with
the hard-negative term, and the rollback + commit double pass of ``forward``.

What runs where: backbones -> native iResNet executor (model/iresnet.py); pool bookkeeping ->
native LRU (lru.py / head.py); both pool contractions, margins, cross-entropy, top-k and dL/dp ->
the fused head kernels (head.py); EMA -> one fused pass over the parameters.
"""
import ctypes

import torch
import torch.nn.functional as F
from torch.nn import Module

from . import _lib
from .head import DcpHead, QueuePositionView
from .model import create_net


POOL_CHUNK = 1 << 16          # slots per generator chunk of a device-built pool
POOL_HOST_LIMIT = 256 << 20   # pools up to this many bytes are built exactly as the reference does (host, global RNG)


def build_pool(queue_size, feat_dim, device=None, shard=None, seed=None):
    """``F.normalize(torch.rand(2, Q, D), dim=2)`` (ffc.py:29-30).

    Small pools (and ``device='cpu'``) use that very expression, so ``torch.manual_seed`` gives the
    reference's pool.  Large pools (10 M identities: 41 GB; 100 M: 410 GB — SURVEY a1) are never
    materialised on the host: the same distribution is drawn straight into HBM in chunks of POOL_CHUNK
    slots, chunk c from its own generator seeded ``seed + c`` (``seed`` is one draw from the global
    CPU generator, so ``torch.manual_seed`` still fixes the pool and every rank that seeds alike agrees).
    ``shard = (rank, world)`` builds only slots [rank * Q / world, (rank + 1) * Q / world) — bit-identical
    to that slice of the full pool, which is what lets an identity-sharded run (parallel.ShardedFFC)
    start from the single-GPU run's pool without any rank ever holding all of it."""
    lo, n = 0, queue_size
    if shard is not None:
        rank, world = shard
        if queue_size % world:
            raise ValueError("queue_size must be divisible by the number of ranks for a sharded pool")
        n = queue_size // world
        lo = rank * n
    nbytes = 2 * n * feat_dim * 4
    if device is None:
        device = 'cuda' if (torch.cuda.is_available() and nbytes > POOL_HOST_LIMIT) else 'cpu'
    device = torch.device(device)
    if device.type == 'cpu' and shard is None and seed is None:
        return F.normalize(torch.rand(2, queue_size, feat_dim), dim=2)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    gen = torch.Generator(device=device)
    out = torch.empty(2, n, feat_dim, dtype=torch.float32, device=device)
    c = lo // POOL_CHUNK
    while c * POOL_CHUNK < lo + n:
        a, b = c * POOL_CHUNK, min((c + 1) * POOL_CHUNK, queue_size)
        gen.manual_seed(seed + c)
        blk = F.normalize(torch.rand(2, b - a, feat_dim, generator=gen, device=device), dim=2)
        s0, s1 = max(a, lo), min(b, lo + n)
        out[:, s0 - lo:s1 - lo] = blk[:, s0 - a:s1 - a]
        c += 1
    return out


class FFC(Module):
    def __init__(self, net_type, feat_dim, queue_size=7409, scale=32.0, loss_type='AM', margin=0.4, momentum=0.99,
                 neg_margin=0.25, pretrained_model_path=None, num_class=None, precise_head=False,
                 pool_device=None, pool_shard=None):
        super(FFC, self).__init__()
        assert loss_type in ('AM', 'Arc', 'SV')                       # ffc.py:17
        self.device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.probe_net = create_net(net_type, feat_dim=feat_dim, fp16=True)      # ffc.py:22-23
        self.gallery_net = create_net(net_type, feat_dim=feat_dim, fp16=True)
        self.pool_shard = pool_shard
        self.register_buffer('queue', build_pool(queue_size, feat_dim, pool_device, pool_shard))   # ffc.py:29-30
        self.queue_size = queue_size
        self.feat_dim = feat_dim
        self.scale = scale
        self.margin = margin
        self.loss_type = loss_type
        self.neg_margin = neg_margin
        self.register_buffer('mask', torch.zeros(self.queue_size, 1))  # ffc.py:45 (state-dict parity; unused here)
        self.m = momentum
        self.mask_svfc = 1.2
        self.hard_neg = min(max(int(self.queue_size * 0.0002), 3), 10)
        self.precise_head = precise_head
        self._head = None
        self._head_qptr = None
        self.__dict__['_bootstrap_head'] = DcpHeadState(queue_size)
        for param_p, param_g in zip(self.probe_net.parameters(), self.gallery_net.parameters()):   # ffc.py:53-55
            param_g.data.copy_(param_p.data)
            param_g.requires_grad = False

    # -- reference attributes read by main.py:85 ----------------------------------------------------
    @property
    def lru(self):
        return self._state().lru

    @property
    def queue_position_dict(self):
        return QueuePositionView(self._state().qp)

    def _state(self):
        return self._head if self._head is not None else self.__dict__['_bootstrap_head']

    def _ensure_head(self):
        """(Re)binds the native head to the device pool tensor (after .cuda() / load_state_dict)."""
        q = self.queue
        if not q.is_cuda:
            raise _lib.VlsfrError("FFC.forward: move the module to the GPU first (.cuda()); there is no CPU path")
        if q.shape[1] != self.queue_size:
            raise _lib.VlsfrError("this FFC holds pool slots of shard %r only: drive it through parallel.ShardedFFC" %
                                  (self.pool_shard,))
        if self._head is None or self._head_qptr != q.data_ptr():
            if not q.is_contiguous():
                self.queue = q = q.contiguous()
            old = self._state()
            head = DcpHead(q, self.scale, self.margin, self.loss_type, precise=self.precise_head)
            head.lru, head.qp = old.lru, old.qp          # keep the allocator state across re-binds
            self._head, self._head_qptr = head, q.data_ptr()
        head = self._head
        # the reference reads these attributes on every add_margin call (ffc.py:60-138): follow later changes
        head.scale, head.margin, head.loss_type, head.precise = float(self.scale), float(self.margin), self.loss_type, bool(self.precise_head)
        head.hard_neg = int(self.hard_neg)
        if self.__dict__.get('head_dtype'):          # 'bf16' | 'fp8' (e4m3 sweep, config C5's precision); default: VLSFR_HEAD_DTYPE or bf16
            head.head_dtype = self.__dict__['head_dtype']
        return head

    @torch.no_grad()
    def _momentum_update_gallery(self):                               # ffc.py:139-145
        from .optim.fused import ema_update
        ema_update(list(self.gallery_net.parameters()), list(self.probe_net.parameters()), self.m)
        self.gallery_net.weights_dirty = True

    def embed_pair(self, p_data, g_data, update_gallery):
        """p = probe_net(p_data) (autograd) and g = gallery_net(g_data) (no grad), the two backbones of one
        pass (ffc.py:156-161 / :211-217).  They share nothing but the input batch, so the gallery pass runs
        on a second HIP stream beside the probe pass: the tail of every kernel of one net (a 14x14 layer
        fills 1.5 waves of workgroups) is covered by the kernels of the other.  The EMA of the rollback pass
        reads only probe PARAMETERS, which the probe forward does not change, so doing it first is the
        reference order's result."""
        if not self.__dict__.get('concurrent_streams', True):        # A/B switch (bench.py --serial)
            with torch.no_grad():
                if update_gallery:
                    self._momentum_update_gallery()
                g = self.gallery_net(g_data)
            return self.probe_net(p_data), g
        main = torch.cuda.current_stream()
        side = self.__dict__.get('_side_stream')
        if side is None or side.device != main.device:
            side = torch.cuda.Stream(device=main.device)
            self.__dict__['_side_stream'] = side
        with torch.no_grad():
            if update_gallery:
                self._momentum_update_gallery()
            side.wait_stream(main)                       # inputs and the EMA'd weights are ready
            with torch.cuda.stream(side):
                g = self.gallery_net(g_data)
        p = self.probe_net(p_data)
        main.wait_stream(side)
        g.record_stream(main)
        return p, g

    def embed_both(self, x, y):
        """The four backbone passes of a step — probe(x), gallery(y) of the rollback pass and probe(y), gallery(x) of the
        committing pass (ffc.py:264-267 -> :211-217, :156-161) — on four HIP streams.  They depend on nothing but the inputs
        and the weights (the EMA of ffc.py:139-145 runs once, before either gallery pass, in both orders of execution), so
        issuing them together changes no result except the ORDER of the two running-statistics updates of every BatchNorm,
        which is kept by deferring them (NativeBackbone.begin_deferred_running / merge_deferred_running, called by
        finish_both).  The idea: a chain alternates MFMA-bound convolutions with HBM-bound normalisation kernels and partial
        last waves, so more chains should keep a convolution resident more of the time.  MEASURED (ir100, 10 M identities,
        batch 256, same box, alternating): forward phase 50.9 - 51.1 ms with four chains against 48.3 - 48.6 ms with two,
        93.5 against 90.3 ms per step — four MFMA-bound kernels side by side cost more in the shared L2 than the gaps they
        fill, the same outcome as the weight gradients on a fourth stream.  Kept as an option
        (`forward_chains = 4`, bench.py --fwd-chains 4) with its parity test; the default stays two chains per pass.
        Returns ((p1, g1, ready1), (p2, g2, ready2)): ready_k = the events a consumer stream waits on for pass k."""
        main = torch.cuda.current_stream()
        dev = main.device
        ch = self.__dict__.get('_chains')
        if ch is None or ch[0].device != dev:
            ch = self.__dict__['_chains'] = [torch.cuda.Stream(device=dev) for _ in range(4)]
        B = int(x.shape[0])
        with torch.no_grad():
            self._momentum_update_gallery()
            self.gallery_net.prepare_weights(B, dev)
        self.probe_net.prepare_weights(B, dev)
        self.probe_net.begin_deferred_running()
        self.gallery_net.begin_deferred_running()
        for s in ch:
            s.wait_stream(main)
        p1, e_p1 = self.probe_net.run_chain(x, 0, ch[0])
        with torch.no_grad():
            g1, e_g1 = self.gallery_net.run_chain(y, 0, ch[1])
        p2, e_p2 = self.probe_net.run_chain(y, 1, ch[2])
        with torch.no_grad():
            g2, e_g2 = self.gallery_net.run_chain(x, 1, ch[3])
        return (p1, g1, (e_p1, e_g1)), (p2, g2, (e_p2, e_g2))

    def finish_both(self):
        """After the consumers of embed_both have been joined into the current stream: both running-statistics updates
        of every BatchNorm, in pass order."""
        self.probe_net.merge_deferred_running()          # the current stream already waits for the four passes (run_chain)
        self.gallery_net.merge_deferred_running()

    def forward_impl(self, p_data, g_data, probe_label, gallery_label):          # ffc.py:153-204
        head = self._ensure_head()
        p, g = self.embed_pair(p_data, g_data, update_gallery=False)
        return head.run_pass(p, g, probe_label, gallery_label, transactional=False)

    def forward_impl_rollback(self, p_data, g_data, probe_label, gallery_label):  # ffc.py:208-260
        head = self._ensure_head()
        p, g = self.embed_pair(p_data, g_data, update_gallery=True)
        return head.run_pass(p, g, probe_label, gallery_label, transactional=True)

    def forward(self, x, y, x_label, y_label):                        # ffc.py:264-267
        if not self.__dict__.get('concurrent_streams', True) or not x.is_cuda:
            loss2 = self.forward_impl_rollback(x, y, x_label, y_label)
            loss1 = self.forward_impl(y, x, y_label, x_label)
            return loss1 + loss2
        # Same two passes, with the head of the rollback pass on a third stream beside the backbones of the
        # commit pass (the head needs only p, g of ITS pass; pool and LRU state are touched in program order:
        # both heads run on the head stream, the host bookkeeping is synchronous).
        head = self._ensure_head()
        main = torch.cuda.current_stream()
        hs = self.__dict__.get('_head_stream')
        if hs is None or hs.device != main.device:
            hs = torch.cuda.Stream(device=main.device)
            self.__dict__['_head_stream'] = hs
        marks = self.__dict__.get('_marks')          # diagnostic: bench.py --phases (main-stream events)

        def mark(name):
            if marks is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append((name, e))

        mark("start")
        if self.__dict__.get('forward_chains', 2) == 4 and hasattr(self.probe_net, 'run_chain'):   # measured slower (below): opt-in
            (p1, g1, r1), (p2, g2, r2) = self.embed_both(x, y)
            for e in r1:
                hs.wait_event(e)
            with torch.cuda.stream(hs):
                loss2 = head.run_pass(p1, g1, x_label, y_label, transactional=True)
                for e in r2:
                    hs.wait_event(e)
                loss1 = head.run_pass(p2, g2, y_label, x_label, transactional=False)
                total = loss1 + loss2
            for t in (p1, g1, p2, g2):
                t.record_stream(hs)
            main.wait_stream(hs)
            self.finish_both()
            mark("head 2")
            total.record_stream(main)
            return total
        p1, g1 = self.embed_pair(x, y, update_gallery=True)
        mark("backbones of pass 1")
        hs.wait_stream(main)
        with torch.cuda.stream(hs):
            loss2 = head.run_pass(p1, g1, x_label, y_label, transactional=True)
        p1.record_stream(hs)
        g1.record_stream(hs)
        p2, g2 = self.embed_pair(y, x, update_gallery=False)
        mark("backbones of pass 2 (head 1 beside them)")
        hs.wait_stream(main)
        with torch.cuda.stream(hs):
            loss1 = head.run_pass(p2, g2, y_label, x_label, transactional=False)
            total = loss1 + loss2
        p2.record_stream(hs)
        g2.record_stream(hs)
        main.wait_stream(hs)
        mark("head 2")
        total.record_stream(main)
        return total


class DcpHeadState(object):
    """Allocator state that exists before the pool tensor reaches the device."""

    def __init__(self, queue_size):
        import numpy as np
        from .lru import LRU
        self.lru = LRU(queue_size)
        self.qp = np.zeros(queue_size, dtype=np.uint8)
