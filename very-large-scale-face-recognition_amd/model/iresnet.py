"""iResNet backbone — same module tree / state-dict keys / initialisation as the reference
(model/resnet_arcface.py:26-55 IBasicBlock, :58-152 IResNet, :162-184 factories), executed by the
native gfx950 executor (csrc/iresnet.cpp) instead of per-layer PyTorch ops.

The torch.nn layers below are parameter containers only — they are never called.  Convolution
weights live in channels_last memory ([Cout][R][S][Cin], what the MFMA kernels read); their logical
shape and state-dict layout stay OIHW, so checkpoints interchange with the reference.
"""
import ctypes

import torch
from torch import nn

from .. import _lib


class IBasicBlock(nn.Module):           # resnet_arcface.py:26-55
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(IBasicBlock, self).__init__()
        self.bn1 = nn.BatchNorm2d(inplanes, eps=1e-05)
        self.conv1 = nn.Conv2d(inplanes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, eps=1e-05)
        self.prelu = nn.PReLU(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes, eps=1e-05)
        self.downsample = downsample
        self.stride = stride


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _BackboneFn(torch.autograd.Function):
    """One node for the whole backbone: forward = vlsfr_iresnet_forward, backward =
    vlsfr_iresnet_backward, which accumulates straight into the parameters' .grad buffers."""

    @staticmethod
    def forward(ctx, x, net, *params):
        emb, ws = net._run_forward(x, save=True)
        ctx.net, ctx.ws, ctx.B = net, ws, x.shape[0]
        return emb

    @staticmethod
    def backward(ctx, demb):
        ctx.net._run_backward(demb.contiguous().float(), ctx.ws, ctx.B)
        ctx.ws = None
        return (None, None) + (None,) * len(ctx.net._plist)


class IResNet(nn.Module):
    fc_scale = 7 * 7

    def __init__(self, layers, dropout=0, feat_dim=512, fp16=False, image_size=112):
        super(IResNet, self).__init__()
        if dropout != 0:
            raise _lib.VlsfrError("IResNet: only dropout=0 (the reference default, resnet_arcface.py:61) is covered")
        self.fp16 = fp16                      # kept for API parity; the executor always computes in bf16/fp32
        self.layers_cfg = tuple(int(v) for v in layers)
        self.feat_dim, self.image_size = int(feat_dim), int(image_size)
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64, eps=1e-05)
        self.prelu = nn.PReLU(64)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1])
        self.layer3 = self._make_layer(256, layers[2])
        self.layer4 = self._make_layer(512, layers[3])
        self.bn2 = nn.BatchNorm2d(512, eps=1e-05)
        self.dropout = nn.Dropout(p=dropout, inplace=True)
        side = image_size // 16
        self.fc = nn.Linear(512 * side * side, feat_dim)
        self.features = nn.BatchNorm1d(feat_dim, eps=1e-05)
        nn.init.constant_(self.features.weight, 1.0)          # resnet_arcface.py:97-98
        self.features.weight.requires_grad = False
        for m in self.modules():                               # resnet_arcface.py:100-105
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.1)
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._handles = {}
        self._wcache = None
        self._w_sig = None
        self._scratch = None
        self._eval_ctx = None
        self._nbt_pending = 0
        self.weights_dirty = True

    def flush_counters(self):
        """Adds the forward passes seen since the last flush to every BatchNorm's num_batches_tracked
        (kept off the per-step path: the reference bumps ~80 tiny tensors per forward)."""
        if self._nbt_pending:
            for name, b in self.named_buffers():
                if name.endswith("num_batches_tracked"):
                    b += self._nbt_pending
            self._nbt_pending = 0

    def state_dict(self, *args, **kwargs):
        self.flush_counters()
        return super(IResNet, self).state_dict(*args, **kwargs)

    def _make_layer(self, planes, blocks):                     # resnet_arcface.py:112-136 (stride 2, no dilation)
        downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, 2, bias=False),
                                   nn.BatchNorm2d(planes, eps=1e-05))
        layers = [IBasicBlock(self.inplanes, planes, 2, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(IBasicBlock(planes, planes))
        return nn.Sequential(*layers)

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
        return params, running

    def _handle(self, B, device):
        L = _lib.lib()
        key = (B, str(device))
        if key not in self._handles:
            h = ctypes.c_void_p()
            lay = (ctypes.c_int32 * 4)(*self.layers_cfg)
            L.vlsfr_iresnet_create.restype = ctypes.c_int
            _lib.check(L.vlsfr_iresnet_create(lay, ctypes.c_int32(self.feat_dim), ctypes.c_int32(B),
                                              ctypes.c_int32(self.image_size), ctypes.byref(h)), "vlsfr_iresnet_create")
            for fn in ("vlsfr_iresnet_wcache_bytes", "vlsfr_iresnet_ctx_bytes", "vlsfr_iresnet_scratch_bytes"):
                getattr(L, fn).restype = ctypes.c_size_t
                getattr(L, fn).argtypes = [ctypes.c_void_p]
            L.vlsfr_iresnet_num_params.restype = ctypes.c_int32
            L.vlsfr_iresnet_num_params.argtypes = [ctypes.c_void_p]
            L.vlsfr_iresnet_num_bn.argtypes = [ctypes.c_void_p]
            assert L.vlsfr_iresnet_num_params(h) == len(self._plist)
            sizes = (L.vlsfr_iresnet_wcache_bytes(h), L.vlsfr_iresnet_ctx_bytes(h), L.vlsfr_iresnet_scratch_bytes(h))
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
            L.vlsfr_iresnet_prepare_weights.restype = ctypes.c_int
            _lib.check(L.vlsfr_iresnet_prepare_weights(h, _ptr_array(params), ctypes.c_void_p(self._wcache.data_ptr()),
                                                       _stream()), "vlsfr_iresnet_prepare_weights")
            self._w_sig, self.weights_dirty = sig, False
        if self._scratch is None or self._scratch.numel() < sizes[2] or self._scratch.device != device:
            self._scratch = torch.empty(sizes[2], dtype=torch.uint8, device=device)

    def _run_forward(self, x, save):
        if not x.is_cuda:
            raise _lib.VlsfrError("IResNet.forward needs a device tensor: the backbone has no CPU path")
        L = _lib.lib()
        B = int(x.shape[0])
        assert tuple(x.shape[1:]) == (3, self.image_size, self.image_size), x.shape
        x = x.contiguous().float()
        h, sizes = self._handle(B, x.device)
        params, running = self._tables()
        self._prepare(h, sizes, params, x.device)
        if save:
            ws = torch.empty(sizes[1], dtype=torch.uint8, device=x.device)
        else:
            if self._eval_ctx is None or self._eval_ctx.numel() < sizes[1] or self._eval_ctx.device != x.device:
                self._eval_ctx = torch.empty(sizes[1], dtype=torch.uint8, device=x.device)
            ws = self._eval_ctx
        emb = torch.empty(B, self.feat_dim, dtype=torch.float32, device=x.device)
        L.vlsfr_iresnet_forward.restype = ctypes.c_int
        run_tab = _ptr_array(running) if self.training else None
        _lib.check(L.vlsfr_iresnet_forward(h, ctypes.c_void_p(x.data_ptr()), _ptr_array(params), run_tab,
                                           ctypes.c_void_p(self._wcache.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                           ctypes.c_void_p(self._scratch.data_ptr()), ctypes.c_void_p(emb.data_ptr()),
                                           _stream()), "vlsfr_iresnet_forward")
        if self.training:
            self._nbt_pending += 1          # num_batches_tracked is materialised lazily (flush_counters)
        self._keep = x
        return emb, ws

    def _run_backward(self, demb, ws, B):
        L = _lib.lib()
        h, sizes = self._handle(B, demb.device)
        params, _ = self._tables()
        grads = []
        for p in params:
            if not p.requires_grad:
                grads.append(None)
                continue
            if p.grad is None:
                p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif p.dim() == 4 and not p.grad.permute(0, 2, 3, 1).is_contiguous():
                p.grad = p.grad.contiguous(memory_format=torch.channels_last)
            grads.append(p.grad)
        L.vlsfr_iresnet_backward.restype = ctypes.c_int
        _lib.check(L.vlsfr_iresnet_backward(h, ctypes.c_void_p(demb.data_ptr()), _ptr_array(params), _ptr_array(grads),
                                            ctypes.c_void_p(self._wcache.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                            ctypes.c_void_p(self._scratch.data_ptr()), _stream()),
                   "vlsfr_iresnet_backward")

    def forward(self, x):
        # Training mode always (the reference never calls .eval(), ffc.py:22-23); eval() only stops
        # the running-statistics update.
        params = self._plist
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _BackboneFn.apply(x, self, *params)
        return self._run_forward(x, save=False)[0]

    def __del__(self):
        try:
            L = _lib.lib()
            for h, _ in self._handles.values():
                L.vlsfr_iresnet_destroy(h)
        except Exception:
            pass


def _iresnet(layers, **kwargs):
    kwargs.pop("pretrained", None)
    kwargs.pop("progress", None)
    return IResNet(layers, **kwargs)


def iresnet18(**kw):
    return _iresnet([2, 2, 2, 2], **kw)


def iresnet34(**kw):
    return _iresnet([3, 4, 6, 3], **kw)


def iresnet50(**kw):
    return _iresnet([3, 4, 14, 3], **kw)


def iresnet100(**kw):
    return _iresnet([3, 13, 30, 3], **kw)


def iresnet200(**kw):
    return _iresnet([6, 26, 60, 6], **kw)
