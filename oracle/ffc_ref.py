"""ORACLE — test infrastructure only (never imported by the product package).

CPU restatement (plain PyTorch CPU ops) of the FFC Dynamic-Class-Pool training step:
reference ffc.py:10-267 (FFC), with the three facts the reference leaves implicit made explicit
(SURVEY.md F5-F8):
  * device-agnostic (the reference hard-codes .cuda());
  * snapshot semantics — dL/dp is taken against the pool as it stood at contraction time
    (under CUDA autocast the reference's fp16 cast makes that copy implicitly);
  * no clamp inside the ArcFace square root (ffc.py:101), duplicate (row, slot) writes resolved
    "highest batch index wins".

Pinned against the reference itself: tests/golden/head_*.npz and step_*.npz were produced by
importing the reference ffc.py in the build container with the two harness shims of SURVEY §8(c)
(tests/golden/make_golden.py); tests/test_oracle_golden.py replays them through this module.
"""
import math

import torch
import torch.nn.functional as F

from .lru_ref import LRURef
from . import backbones_ref as bb


def hard_neg_count(queue_size):
    return min(max(int(queue_size * 0.0002), 3), 10)  # ffc.py:48


def dcp_assign_ref(lru, qp, gallery_label, probe_label, transactional):
    """ffc.py:162-177 + 189-192 (commit) / ffc.py:214-235 + 242-245 (rollback)."""
    rows, cols, ones, saved = [], [], [], {}
    for gl in gallery_label:
        known = gl in lru
        idx = lru.try_get(gl) if transactional else lru.get(gl)
        if transactional and idx not in saved:
            saved[idx] = qp[idx]
        if not known:
            rows.append(0)
            qp[idx] = 1
        else:
            rows.append(qp[idx])
            if idx not in ones:
                ones.append(idx)
            qp[idx] = (qp[idx] + 1) % 2
        cols.append(idx)
    labels = [lru.view(pl) for pl in probe_label]
    return rows, cols, labels, ones, saved


def margin_loss_ref(cos, label, loss_type, scale, margin, hard_neg, mask_svfc=1.2):
    """ffc.py:60-138 (add_margin), out of place."""
    pos = label != -1
    out = ~pos
    loss = cos.new_zeros(())
    if pos.any():
        c = cos[pos]
        t = label[pos].view(-1, 1)
        gt = c.gather(1, t)
        if loss_type == "AM":
            new = gt - margin                                                   # :81
        elif loss_type == "Arc":
            new = gt * math.cos(margin) - torch.sqrt(1.0 - gt * gt) * math.sin(margin)   # :100-102
        else:
            hard = c > (gt - margin)                                            # :122
            c = torch.where(hard, mask_svfc * c + mask_svfc - 1.0, c)           # :124-125
            new = torch.where(gt > margin, gt - margin, gt)                     # :123
        c = c.scatter(1, t, new)
        loss = loss + F.cross_entropy(c * scale, t.view(-1))                   # :83 / :104 / :127
    if out.any():
        oc = cos[out]
        k = min(hard_neg, oc.shape[1])
        top = torch.topk(oc, k, dim=1).values                                   # :88-89 (first k of a descending sort)
        loss = loss + torch.clamp(top, min=0).mean()                           # :89-90
    return loss


def head_pass_ref(queue, lru, qp, p, g, probe_label, gallery_label, transactional, loss_type, scale, margin,
                  hard_neg):
    """One forward_impl (transactional=False, ffc.py:153-204) or forward_impl_rollback
    (transactional=True, ffc.py:208-260) given the embeddings.  `queue` [2,Q,D] is updated in place
    only by the committing pass."""
    rows, cols, labels, ones, saved = dcp_assign_ref(lru, qp, [int(v) for v in gallery_label],
                                                     [int(v) for v in probe_label], transactional)
    W = queue.detach().clone()
    gd = g.detach().to(W.dtype)
    for i, (r, c) in enumerate(zip(rows, cols)):                                # :182 / :241, last writer wins
        W[r, c] = gd[i]
    w1 = W[0]
    w2 = W[0].clone()
    if ones:
        oi = torch.tensor(ones, dtype=torch.long)
        w2[oi] = W[1][oi]                                                       # :198-200 / :250-252
    label = torch.tensor(labels, dtype=torch.long)
    cos1 = p @ w1.t()                                                           # :195 / :248
    cos2 = p @ w2.t()                                                           # :201 / :253
    loss = (margin_loss_ref(cos1, label, loss_type, scale, margin, hard_neg)
            + margin_loss_ref(cos2, label, loss_type, scale, margin, hard_neg))
    if transactional:
        for k, v in saved.items():                                              # :256-257
            qp[k] = v
        lru.rollback_steps(len(rows))                                           # :259
    else:
        queue.copy_(W)
    return loss, dict(rows=rows, cols=cols, labels=labels, ones=ones)


class FFCRef(object):
    """Whole-step restatement: FFC.__init__ (ffc.py:11-55) + forward (ffc.py:264-267)."""

    def __init__(self, net_type, feat_dim, queue_size, scale=32.0, loss_type="AM", margin=0.4, momentum=0.99,
                 gen=None, layers=None, dtype=torch.float32, emulate_bf16=False):
        assert loss_type in ("AM", "Arc", "SV")
        self.probe, self.fwd = bb.make_backbone(net_type, feat_dim, gen, layers, emulate_bf16)
        self.net_type = net_type
        self.probe = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in self.probe.items()}
        for k, v in self.probe.items():
            if bb.trainable(k, net_type):
                v.requires_grad_(True)
        # ffc.py:53-55: parameters are copied p -> g and frozen; gallery buffers keep their own defaults
        self.gallery = {k: v.detach().clone() for k, v in self.probe.items()}
        q = torch.rand(2, queue_size, feat_dim, generator=gen).to(dtype)
        self.queue = F.normalize(q, dim=2)                                       # ffc.py:29-30
        self.lru = LRURef(queue_size)
        self.qp = [0] * queue_size                                               # ffc.py:41-43
        self.scale, self.margin, self.loss_type, self.m = scale, margin, loss_type, momentum
        self.hard_neg = hard_neg_count(queue_size)

    def parameters(self):
        return [v for k, v in self.probe.items() if bb.trainable(k, self.net_type)]

    def ema(self):                                                               # ffc.py:139-145 (parameters only,
        with torch.no_grad():                                                    # frozen features.weight included)
            for k, v in self.probe.items():
                if not bb.is_buffer(k):
                    self.gallery[k] = self.gallery[k] * self.m + v.detach() * (1.0 - self.m)

    def _head(self, p, g, pl, gl, transactional):
        return head_pass_ref(self.queue, self.lru, self.qp, p, g, pl, gl, transactional, self.loss_type,
                             self.scale, self.margin, self.hard_neg)[0]

    def forward(self, x, y, x_label, y_label):
        p = self.fwd(self.probe, x)                                              # ffc.py:209
        self.ema()                                                               # :211
        with torch.no_grad():
            g = self.fwd(self.gallery, y)                                        # :212
        loss2 = self._head(p, g, x_label.tolist(), y_label.tolist(), True)
        p = self.fwd(self.probe, y)                                              # :157
        with torch.no_grad():
            g = self.fwd(self.gallery, x)                                        # :159
        loss1 = self._head(p, g, y_label.tolist(), x_label.tolist(), False)
        return loss1 + loss2


def sgd_nesterov_step_ref(params, grads, bufs, lr, momentum=0.9, weight_decay=1e-4, nesterov=True):
    """torch.optim.SGD update rule with momentum and weight decay, nesterov by default (reference
    optim/optimizer.py:148-150 builds it from config/optim_config:9-13).  bufs[i] is None on the first step."""
    with torch.no_grad():
        for i, (p, g) in enumerate(zip(params, grads)):
            d = g + weight_decay * p
            if bufs[i] is None:
                bufs[i] = d.clone()
            else:
                bufs[i].mul_(momentum).add_(d)
            p.add_(d + momentum * bufs[i] if nesterov else bufs[i], alpha=-lr)
    return bufs
