"""Shared machinery of the natively executed backbones (iResNet, MobileFaceNet): the torch.nn layers of
a backbone are parameter containers only; forward / backward are single calls into the C++ executor
(csrc/iresnet.cpp, csrc/mobilenet.cpp), which accumulates gradients straight into the parameters'
.grad buffers."""
import contextlib
import ctypes
import weakref

import torch
from torch import nn

from .. import _lib


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _BackboneFn(torch.autograd.Function):
    """One node for the whole backbone: forward = <prefix>_forward, backward = <prefix>_backward, which
    accumulates straight into the parameters' .grad buffers (atomically, so that the backward passes of
    the two forward passes of an FFC step can run side by side: every second one goes to a second HIP
    stream with its own scratch, and the autograd engine's end-of-backward callback joins it)."""

    @staticmethod
    def forward(ctx, x, net, *params):
        ctx.slot = net._fwd_slot
        net._fwd_slot ^= 1
        emb, ws = net._run_forward(x, save=True, owner=ctx, slot=ctx.slot, chain=net._chain)
        ctx.net, ctx.ws, ctx.B = net, ws, x.shape[0]
        return emb

    @staticmethod
    def backward(ctx, demb):
        net = ctx.net
        demb = demb.contiguous().float()
        if ctx.slot == 1 and net.concurrent_backward:
            main = torch.cuda.current_stream()
            side = net._side_stream(main.device)
            net._ensure_grads()                         # allocate / zero missing .grad on the main stream
            side.wait_stream(main)                      # demb and everything before it
            with torch.cuda.stream(side):
                net._run_backward(demb, ctx.ws, ctx.B, alt=True, slot=1)
            demb.record_stream(side)
            ctx.ws.record_stream(side)
            if not net._join_queued:
                net._join_queued = True

                def join():
                    net._join_queued = False
                    torch.cuda.current_stream().wait_stream(side)
                torch.autograd.Variable._execution_engine.queue_callback(join)
        else:
            net._run_backward(demb, ctx.ws, ctx.B, slot=ctx.slot)
        ctx.ws = None
        return (None, None) + (None,) * len(net._plist)


class NativeBackbone(nn.Module):
    """Subclasses define `_cprefix` ("vlsfr_iresnet" / "vlsfr_mobilenet"), `_create_args()` (the
    ctypes arguments of <prefix>_create before the batch size... see _handle) and the module tree."""
    _cprefix = None
    feat_dim = None
    image_size = 112
    overlap_wgrad = False      # True: executors that export <prefix>_backward_overlap run the weight gradients on a side stream
                               # (measured at ir100 / batch 256: 94.0 vs 91.6 ms per step — a third and fourth MFMA-bound kernel
                               # beside the two backward chains cost more in shared LDS / L2 than the BatchNorm gaps they fill)

    def _init_native(self):
        self._handles = {}
        self._wcache = None
        self._w_sig = None
        self._scratch = None
        self._scratch_alt = None
        self._ctx_pool = [None, None]      # saved-activation buffers of the two passes of a step, reused across steps
        self._ctx_owner = [None, None]     # weakref to the autograd ctx that still needs the buffer
        self._eval_ctx = [None, None]      # saved-activation buffers of the no-grad passes (one per concurrent chain)
        self._chain = None                 # (stream, pass index) while run_chain() is on the stack
        self.use_graphs = False            # True: forward / backward executor calls replayed from HIP graphs (launch-bound sizes)
        self._graphs = {}
        self._deferred = None              # zeroed stand-ins for the running statistics (begin_deferred_running)
        self._fwd_slot = 0
        self._join_queued = False
        self._bwd_stream = None
        self.concurrent_backward = True
        self._nbt_pending = 0
        self.weights_dirty = True

    def flush_counters(self):
        """Adds the forward passes seen since the last flush to every BatchNorm's num_batches_tracked
        (kept off the per-step path: the reference bumps one tiny tensor per BatchNorm per forward)."""
        if self._nbt_pending:
            for name, b in self.named_buffers():
                if name.endswith("num_batches_tracked"):
                    b += self._nbt_pending
            self._nbt_pending = 0

    def state_dict(self, *args, **kwargs):
        self.flush_counters()
        return super(NativeBackbone, self).state_dict(*args, **kwargs)

    def _create(self, L, B, h):
        raise NotImplementedError

    def _side_stream(self, device):
        if self._bwd_stream is None or self._bwd_stream.device != device:
            self._bwd_stream = torch.cuda.Stream(device=device)
        return self._bwd_stream

    # ------------------------------------------------------------------------------------------
    @property
    def _plist(self):
        cache = self.__dict__.get("_plist_cache")
        if cache is None:
            cache = [p for _, p in self.named_parameters()]
            self.__dict__["_plist_cache"] = cache
        return cache

    def _tables(self):
        """Pointer tables in the executor's order (= registration order of the reference module)."""
        params = self._plist
        # The layout check walks every parameter (a permuted view each) and every buffer: ~1 ms of host time per call for
        # ir100, paid six times a step.  It is repeated only when the parameter storage has visibly changed (first / last
        # pointer: .cuda(), .to(), an optimizer re-pointing the parameters into a flat buffer) or after _apply /
        # load_state_dict (hooks below); in-place writes (copy_, optimizer steps) keep strides and need no re-check.
        key = (params[0].data_ptr(), params[-1].data_ptr(), len(params)) if params else None
        cache = self.__dict__.get("_tables_cache")
        if cache is not None and cache[0] == key:
            return params, cache[1]
        for p in params:
            p._vlsfr_owner = self
            if p.dim() == 4 and not p.data.permute(0, 2, 3, 1).is_contiguous():
                p.data = p.data.contiguous(memory_format=torch.channels_last)   # e.g. after load_state_dict copies
            elif p.dim() != 4 and not p.data.is_contiguous():
                p.data = p.data.contiguous()
        running = []
        for name, b in self.named_buffers():
            if name.endswith("running_mean") or name.endswith("running_var"):
                running.append(b)
        key = (params[0].data_ptr(), params[-1].data_ptr(), len(params)) if params else None
        self.__dict__["_tables_cache"] = (key, running)
        return params, running

    def _apply(self, fn, *args, **kwargs):
        self.__dict__["_tables_cache"] = None
        self.__dict__["_plist_cache"] = None
        return super(NativeBackbone, self)._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.__dict__["_tables_cache"] = None
        return super(NativeBackbone, self).load_state_dict(*args, **kwargs)

    def _handle(self, B, device):
        L = _lib.lib()
        key = (B, str(device))
        if key not in self._handles:
            h = ctypes.c_void_p()
            self._create(L, B, h)
            for fn in ("_wcache_bytes", "_ctx_bytes", "_scratch_bytes"):
                getattr(L, self._cprefix + fn).restype = ctypes.c_size_t
                getattr(L, self._cprefix + fn).argtypes = [ctypes.c_void_p]
            nump = getattr(L, self._cprefix + "_num_params")
            nump.restype = ctypes.c_int32
            nump.argtypes = [ctypes.c_void_p]
            assert nump(h) == len(self._plist), (nump(h), len(self._plist))
            sizes = tuple(getattr(L, self._cprefix + fn)(h) for fn in ("_wcache_bytes", "_ctx_bytes", "_scratch_bytes"))
            self._handles[key] = (h, sizes)
        return self._handles[key]

    def _prepare(self, h, sizes, params, device):
        """bf16 operand copies of the weights, refreshed only when the weights changed."""
        L = _lib.lib()
        sig = (tuple(p.data_ptr() for p in params), tuple(p._version for p in params))
        if self._wcache is None or self._wcache.numel() != sizes[0] or self._wcache.device != device:
            self._wcache = torch.empty(sizes[0], dtype=torch.uint8, device=device)
            self.weights_dirty = True
        if self.weights_dirty or sig != self._w_sig:
            fn = getattr(L, self._cprefix + "_prepare_weights")
            fn.restype = ctypes.c_int
            _lib.check(fn(h, _ptr_array(params), ctypes.c_void_p(self._wcache.data_ptr()), _stream()),
                       self._cprefix + "_prepare_weights")
            self._w_sig, self.weights_dirty = sig, False
        if self._scratch is None or self._scratch.numel() < sizes[2] or self._scratch.device != device:
            self._scratch = torch.empty(sizes[2], dtype=torch.uint8, device=device)

    def _ctx_buffer(self, nbytes, device, owner, slot):
        """Saved-activation buffer for one training forward: the two passes of a step alternate between two
        persistent buffers (multi-GB allocations per step would go through the caching allocator, whose
        occasional hipMalloc / hipFree under stream-recorded blocks stalls the queue); a buffer whose
        previous owner has not run backward yet is not reused -- that pass gets a fresh allocation."""
        prev = self._ctx_owner[slot]() if self._ctx_owner[slot] is not None else None
        free = prev is None or getattr(prev, "ws", None) is None
        buf = self._ctx_pool[slot]
        if not free:
            return torch.empty(nbytes, dtype=torch.uint8, device=device)
        if buf is None or buf.numel() < nbytes or buf.device != device:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._ctx_pool[slot] = buf
        self._ctx_owner[slot] = weakref.ref(owner) if owner is not None else None
        return buf

    def _scratch_for(self, k, nbytes, device):
        """Executor scratch of chain k (0: self._scratch, 1: self._scratch_alt, the one the side-stream backward uses)."""
        if k == 0:
            if self._scratch is None or self._scratch.numel() < nbytes or self._scratch.device != device:
                self._scratch = torch.empty(nbytes, dtype=torch.uint8, device=device)
            return self._scratch
        if self._scratch_alt is None or self._scratch_alt.numel() < nbytes or self._scratch_alt.device != device:
            self._scratch_alt = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._scratch_alt

    GRAPH_WARMUP = 2       # eager calls per signature before its graph is captured (lazy buffers exist, LDS attributes are set)

    def _launch(self, key, sig, inp, out, call):
        """Runs call(inp, out) — one executor call: a few hundred dependent kernel launches — on the current stream, or,
        with `use_graphs`, replays it from a HIP graph captured on the third call with the same buffers (sig(): every pointer
        the call bakes into its launches).  MobileFaceNet at batch 32 issues ~1 200 kernels of a few microseconds per step:
        the step is bound by the launch rate, which a graph launch removes.  The input is copied into the
        graph's own buffer; `out` is returned as a copy of the graph's output buffer.  Not to be combined with the measurement
        hooks (they bracket launches with events on the launching stream: bench.py turns graphs off for its profiled repeats).
        The executors clear their accumulator regions with a zero-fill KERNEL (vlsfr_zero_bytes): with hipMemsetAsync nodes in
        the captured chain, replays gave intermittently wrong BatchNorm statistics on ROCm 7.2 (scripts/graph_debug.py)."""
        if not self.use_graphs:
            call(inp, out)
            return out
        sig = sig()          # (a callable: some six hundred data_ptr() calls that plain launches do not need)
        st = self._graphs.get(key)
        if st is None or st["sig"] != sig:
            st = self._graphs[key] = dict(sig=sig, seen=0, graph=None)
        if st["graph"] is None:
            st["seen"] += 1
            if st["seen"] <= self.GRAPH_WARMUP:
                call(inp, out)
                return out
            st["inp"], st["out"] = torch.empty_like(inp), (torch.empty_like(out) if out is not None else None)
            st["inp"].copy_(inp)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                call(st["inp"], st["out"])
            st["graph"] = g
        st["inp"].copy_(inp)
        st["graph"].replay()
        if out is not None:
            out.copy_(st["out"])
        return out

    def _run_forward(self, x, save, owner=None, slot=0, chain=None):
        """chain = (stream, k): this pass is pass k (0 / 1) of a step whose passes run side by side (FFC.embed_both) — it
        executes on `stream`, with scratch / no-grad context k of its own, and (training) leaves its running-statistics
        contribution in the zeroed table k of begin_deferred_running() instead of updating the buffers."""
        if not x.is_cuda:
            raise _lib.VlsfrError("%s.forward needs a device tensor: the backbone has no CPU path" % type(self).__name__)
        L = _lib.lib()
        B = int(x.shape[0])
        assert tuple(x.shape[1:]) == (3, self.image_size, self.image_size), x.shape
        x = x.contiguous().float()
        h, sizes = self._handle(B, x.device)
        params, running = self._tables()
        k = chain[1] if chain is not None else 0
        with (torch.cuda.stream(chain[0]) if chain is not None else contextlib.nullcontext()):
            self._prepare(h, sizes, params, x.device)
            scratch = self._scratch_for(k, sizes[2], x.device)
            if save:
                ws = self._ctx_buffer(sizes[1], x.device, owner, slot)
            else:
                ws = self._eval_ctx[k]
                if ws is None or ws.numel() < sizes[1] or ws.device != x.device:
                    ws = self._eval_ctx[k] = torch.empty(sizes[1], dtype=torch.uint8, device=x.device)
            emb = torch.empty(B, self.feat_dim, dtype=torch.float32, device=x.device)
            fwd = getattr(L, self._cprefix + "_forward")
            fwd.restype = ctypes.c_int
            fwd_ex = getattr(L, self._cprefix + "_forward_ex", None)      # executors that can skip what only a backward pass reads
            if fwd_ex is not None:
                fwd_ex.restype = ctypes.c_int
            if not self.training:
                run_tab = None
            elif chain is not None:
                if self._deferred is None:
                    raise _lib.VlsfrError("run_chain in training mode needs begin_deferred_running() first")
                run_tab = self._deferred["tabs"][k]
            else:
                run_tab = _ptr_array(running)
            emb = self._launch(("fwd", B, x.device.index, bool(save), slot if save else k, k, run_tab is not None),
                               lambda: (tuple(p.data_ptr() for p in params), None if run_tab is None else tuple(run_tab),
                                        self._wcache.data_ptr(), ws.data_ptr(), scratch.data_ptr()),
                               x, emb,
                               lambda xin, out: _lib.check(
                                   fwd(h, ctypes.c_void_p(xin.data_ptr()), _ptr_array(params), run_tab,
                                       ctypes.c_void_p(self._wcache.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                       ctypes.c_void_p(scratch.data_ptr()), ctypes.c_void_p(out.data_ptr()), _stream())
                                   if fwd_ex is None or save else
                                   fwd_ex(h, ctypes.c_void_p(xin.data_ptr()), _ptr_array(params), run_tab,
                                          ctypes.c_void_p(self._wcache.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                          ctypes.c_void_p(scratch.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_int32(0), _stream()),
                                   self._cprefix + "_forward"))
        if self.training:
            self._nbt_pending += 1          # num_batches_tracked is materialised lazily (flush_counters)
        self.__dict__.setdefault("_keep", {})[k] = x      # the input outlives the asynchronous pass
        return emb, ws

    # ---- the two passes of a step side by side (FFC.embed_both) -------------------------------------------------
    def prepare_weights(self, B, device):
        """bf16 operand copies on the CURRENT stream (before the chains fork from it)."""
        h, sizes = self._handle(B, device)
        params, _ = self._tables()
        self._prepare(h, sizes, params, device)

    def begin_deferred_running(self):
        """Zeroed stand-ins for (running_mean, running_var) of every BatchNorm, one set per pass: the executors' kernels
        compute (1 - m) * 0 + m * s into them, merge_deferred_running() applies both updates in pass order
        (vlsfr_running_merge).  Called on the stream the chains fork from."""
        _, running = self._tables()
        if not running or not self.training:
            self._deferred = None
            return
        key = tuple(b.data_ptr() for b in running)
        d = self._deferred
        if d is None or d["key"] != key:
            pad4 = lambda n: (n + 3) & ~3
            total = sum(pad4(b.numel()) for b in running)
            flat = torch.zeros(2, total, dtype=torch.float32, device=running[0].device)
            views, off = ([], []), 0
            for b in running:
                for k in range(2):
                    views[k].append(flat[k, off:off + b.numel()])
                off += pad4(b.numel())
            rows = [[b.data_ptr(), v0.data_ptr(), v1.data_ptr(), b.numel()] for b, v0, v1 in zip(running, views[0], views[1])]
            import numpy as np
            tab = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(running[0].device)
            d = self._deferred = dict(key=key, flat=flat, views=views, tabs=(_ptr_array(views[0]), _ptr_array(views[1])), merge=tab)
        else:
            d["flat"].zero_()

    def merge_deferred_running(self, momentum=0.1):
        d = self._deferred
        if d is None:
            return
        fn = _lib.lib().vlsfr_running_merge
        fn.restype = ctypes.c_int
        _lib.check(fn(ctypes.c_void_p(d["merge"].data_ptr()), ctypes.c_int32(d["merge"].shape[0]), ctypes.c_float(momentum),
                      _stream()), "vlsfr_running_merge")

    def run_chain(self, x, k, stream):
        """Forward pass k (0 / 1) of the step on `stream` (module call: hooks run).  With autograd the node is created on
        the CALLER's stream (its backward is scheduled exactly like that of a plain forward).  Returns (output, event): the
        event marks the output on `stream`; the caller's stream is also made to wait for it, so code that uses the output
        there (forward hooks, a plain consumer) is ordered after the pass — a consumer that wants to start earlier waits on
        the event from a stream of its own."""
        self._chain = (stream, k)
        try:
            out = self(x)
        finally:
            self._chain = None
        return out, self.__dict__.pop("_chain_event")

    def _ensure_grads(self):
        """.grad buffers in the executor's layout (allocated + zeroed on the CURRENT stream when missing)."""
        grads = []
        for p in self._plist:
            if not p.requires_grad:
                grads.append(None)
                continue
            if p.grad is None:
                p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif p.dim() == 4 and not p.grad.permute(0, 2, 3, 1).is_contiguous():
                p.grad = p.grad.contiguous(memory_format=torch.channels_last)
            grads.append(p.grad)
        return grads

    # ---- gradient buckets for the multi-GPU step (parallel.py): parameter groups whose gradients are complete at
    # the same point of the backward pass, in backward order; a subclass with a staged executor overrides both.
    N_BUCKETS = 1

    def bucket_of(self, name):
        return 0

    def _overlap_state(self, slot, h, device):
        """Ring buffer, events and weight-gradient stream of backward pass `slot` (persistent: the ring is never handed
        back to the allocator, so kernels of the side stream cannot outlive it)."""
        st = self.__dict__.setdefault("_overlap", {})
        L = _lib.lib()
        fn = getattr(L, self._cprefix + "_overlap_ring_bytes")
        fn.restype = ctypes.c_size_t
        need = int(fn(h))
        ov = st.get(slot)
        if ov is None or ov["ring"].numel() < need or ov["ring"].device != device:
            n_ev = int(getattr(L, self._cprefix + "_overlap_events")())
            arr = (ctypes.c_void_p * n_ev)()
            for k in range(n_ev):
                e = ctypes.c_void_p()
                _lib.check(L.vlsfr_event_create(ctypes.byref(e)), "vlsfr_event_create")
                arr[k] = e.value
            ov = dict(ring=torch.empty(need, dtype=torch.uint8, device=device), events=arr, stream=torch.cuda.Stream(device=device))
            st[slot] = ov
        return ov

    def stage_events(self, slot):
        """Events of backward pass `slot` (0 / 1: the two passes of an FFC step), one per bucket, created on first use;
        the pass records event k on ITS stream when bucket k's gradients are enqueued."""
        evs = self.__dict__.setdefault("_stage_events", {})
        if slot not in evs:
            L = _lib.lib()
            arr = (ctypes.c_void_p * self.N_BUCKETS)()
            for k in range(self.N_BUCKETS):
                e = ctypes.c_void_p()
                _lib.check(L.vlsfr_event_create(ctypes.byref(e)), "vlsfr_event_create")
                arr[k] = e.value
            evs[slot] = arr
        return evs[slot]

    def _run_backward(self, demb, ws, B, alt=False, slot=0):
        L = _lib.lib()
        h, sizes = self._handle(B, demb.device)
        scratch = self._scratch
        if alt:                                        # second scratch for the pass on the side stream
            if self._scratch_alt is None or self._scratch_alt.numel() < sizes[2] or self._scratch_alt.device != demb.device:
                self._scratch_alt = torch.empty(sizes[2], dtype=torch.uint8, device=demb.device)
            scratch = self._scratch_alt
        params, _ = self._tables()
        grads = self._ensure_grads()
        signal = self.__dict__.get("signal_stages", False)      # set by parallel.py: somebody waits on the events
        overlap = getattr(L, self._cprefix + "_backward_overlap", None) if self.overlap_wgrad else None
        if overlap is not None:
            # weight gradients on their own stream beside the input-gradient chain of this pass (ring of gradient buffers)
            ov = self._overlap_state(slot, h, demb.device)
            overlap.restype = ctypes.c_int
            _lib.check(overlap(h, ctypes.c_void_p(demb.data_ptr()), _ptr_array(params), _ptr_array(grads),
                               ctypes.c_void_p(self._wcache.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                               ctypes.c_void_p(scratch.data_ptr()), ctypes.c_void_p(ov["ring"].data_ptr()),
                               ctypes.c_size_t(ov["ring"].numel()), self.stage_events(slot) if signal else None,
                               ctypes.c_void_p(ov["stream"].cuda_stream), ov["events"], _stream()),
                       self._cprefix + "_backward_overlap")
            return
        staged = getattr(L, self._cprefix + "_backward_staged", None) if signal and self.N_BUCKETS > 1 else None
        if staged is not None:
            staged.restype = ctypes.c_int
            _lib.check(staged(h, ctypes.c_void_p(demb.data_ptr()), _ptr_array(params), _ptr_array(grads),
                              ctypes.c_void_p(self._wcache.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                              ctypes.c_void_p(scratch.data_ptr()), self.stage_events(slot), _stream()),
                       self._cprefix + "_backward_staged")
            return
        bwd = getattr(L, self._cprefix + "_backward")
        bwd.restype = ctypes.c_int
        if signal:      # an event is recorded right behind the pass: plain launches
            _lib.check(bwd(h, ctypes.c_void_p(demb.data_ptr()), _ptr_array(params), _ptr_array(grads),
                           ctypes.c_void_p(self._wcache.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                           ctypes.c_void_p(scratch.data_ptr()), _stream()), self._cprefix + "_backward")
        else:
            self._launch(("bwd", B, demb.device.index, slot, bool(alt)),
                         lambda: (tuple(p.data_ptr() for p in params), tuple(g.data_ptr() if g is not None else 0 for g in grads),
                                  self._wcache.data_ptr(), ws.data_ptr(), scratch.data_ptr()), demb, None,
                         lambda din, _o: _lib.check(bwd(h, ctypes.c_void_p(din.data_ptr()), _ptr_array(params), _ptr_array(grads),
                                                        ctypes.c_void_p(self._wcache.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                                        ctypes.c_void_p(scratch.data_ptr()), _stream()), self._cprefix + "_backward"))
        if signal:                                              # single bucket: complete when the pass is
            _lib.check(L.vlsfr_event_record(ctypes.c_void_p(self.stage_events(slot)[0]), _stream()), "vlsfr_event_record")

    def forward(self, x):
        # Training mode always (the reference never calls .eval(), ffc.py:22-23); eval() only stops
        # the running-statistics update.
        params = self._plist
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            out = _BackboneFn.apply(x, self, *params)
        else:
            out = self._run_forward(x, save=False, chain=self._chain)[0]
        if self._chain is not None:      # run_chain: order the caller's stream (and with it forward hooks) after the pass
            ev = self.__dict__["_chain_event"] = self._chain[0].record_event()
            torch.cuda.current_stream().wait_event(ev)
        return out

    def __del__(self):
        try:
            L = _lib.lib()
            for h, _ in self._handles.values():
                getattr(L, self._cprefix + "_destroy")(h)
        except Exception:
            pass


