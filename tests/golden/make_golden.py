"""Generates the golden vectors under tests/golden/ by IMPORTING the reference
(/root/reference, read-only) in the build container.  The reference never travels to the GPU box:
only the small .npz / .json outputs of this script are committed.

    python tests/golden/make_golden.py

Harness-side shims (no reference file is modified):
  1. torch.Tensor.cuda -> identity (the reference hard-codes .cuda(), ffc.py:179-180,194,237-238,246);
  2. the name `F` inside the reference's ffc module is replaced by a proxy whose linear() snapshots
     the weight — the copy CUDA autocast makes implicitly.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from tests.golden import common  # noqa: E402
from oracle import backbones_ref as bb  # noqa: E402

import lru as ref_lru  # noqa: E402  (reference)
import ffc as ref_ffc  # noqa: E402  (reference)
from model import resnet_arcface as ref_iresnet  # noqa: E402  (reference)

torch.Tensor.cuda = lambda self, *a, **k: self


class _FProxy(types.ModuleType):
    def __getattr__(self, name):
        return getattr(torch.nn.functional, name)

    @staticmethod
    def linear(x, w, b=None):
        return torch.nn.functional.linear(x, w.clone(), b)


ref_ffc.F = _FProxy("F")
ref_ffc.print = lambda *a, **k: None  # silence the stray print at ffc.py:196


# ------------------------------------------------------------------------------------------------
def lru_traces():
    rng = np.random.default_rng(7)
    cases = []
    for cap, n_keys, n_ops in ((3, 6, 60), (8, 20, 300), (16, 18, 400), (5, 100, 200)):
        l = ref_lru.LRU(cap)
        ops = []
        for _ in range(n_ops):
            depth = len(l.op_stack)
            kind = rng.choice(["get", "try_get", "view", "contains", "rollback", "state"],
                              p=[0.2, 0.35, 0.1, 0.1, 0.15, 0.1])
            key = int(rng.integers(0, n_keys))
            if kind == "get" and depth > 0:
                kind = "try_get"   # committing gets are only issued on an empty undo stack (ffc.py usage)
            if kind == "get":
                ops.append(["get", key, l.get(key)])
            elif kind == "try_get":
                ops.append(["try_get", key, l.try_get(key)])
            elif kind == "view":
                ops.append(["view", key, l.view(key)])
            elif kind == "contains":
                ops.append(["contains", key, int(key in l)])
            elif kind == "rollback":
                steps = int(rng.integers(0, depth + 2))
                l.rollback_steps(steps)
                ops.append(["rollback", steps, len(l.op_stack)])
            else:
                ops.append(["state", 0, [[int(k), int(v)] for k, v in l.state_dict()], l.cur_idx,
                            [o.op_type for o in l.op_stack]])
        ops.append(["state", 0, [[int(k), int(v)] for k, v in l.state_dict()], l.cur_idx,
                    [o.op_type for o in l.op_stack]])
        cases.append(dict(capacity=cap, ops=ops))
    # restore / clear contract
    # (a genuine state_dict always holds the slots 0..n-1: slots are handed out in order, never freed)
    l = ref_lru.LRU(4)
    l.restore([(7, 2), (9, 0), (11, 1)])
    after_restore = [[int(k), int(v)] for k, v in l.state_dict()]
    s = l.get(5)
    s2 = l.get(6)
    state_after_get = [[int(k), int(v)] for k, v in l.state_dict()]
    l.clear()
    extra = dict(after_restore=after_restore, cur_idx_after_restore=3, get5=s, get6=s2,
                 state_after_get=state_after_get, cur_idx_after_clear=l.cur_idx)
    with open(os.path.join(HERE, "lru_traces.json"), "w") as f:
        json.dump(dict(cases=cases, restore_case=extra), f)


# ------------------------------------------------------------------------------------------------
class _Pass(torch.nn.Module):
    def forward(self, x):
        return x


def head_vectors():
    for loss_type, margin in (("AM", 0.4), ("Arc", 0.5), ("SV", 0.35)):
        for tag, (Q, D, B, T, n_id) in (("small", (48, 32, 8, 6, 40)), ("evict", (24, 64, 16, 5, 200)),
                                        ("k10", (51000, 16, 6, 2, 60000))):
            if tag == "k10" and loss_type != "Arc":
                continue
            seed = {"AM": 11, "Arc": 22, "SV": 33}[loss_type] + {"small": 0, "evict": 100, "k10": 200}[tag]
            case = common.head_case(seed, Q, D, B, T, n_id)
            m = ref_ffc.FFC("mobile", D, Q, 32.0, loss_type, margin, 0.99)
            m.probe_net, m.gallery_net = _Pass(), _Pass()
            m.queue = torch.from_numpy(case["queue0"]).clone()
            loss = np.zeros((T, 2), dtype=np.float64)
            dP = np.zeros_like(case["P"])
            for t in range(T):
                xl, yl = torch.from_numpy(case["XL"][t]), torch.from_numpy(case["YL"][t])
                p = torch.from_numpy(case["P"][t, 0]).clone().requires_grad_(True)
                l2 = m.forward_impl_rollback(p, torch.from_numpy(case["G"][t, 0]), xl, yl)
                l2.backward()
                loss[t, 0] = float(l2)
                dP[t, 0] = p.grad.numpy()
                p = torch.from_numpy(case["P"][t, 1]).clone().requires_grad_(True)
                l1 = m.forward_impl(p, torch.from_numpy(case["G"][t, 1]), yl, xl)
                l1.backward()
                loss[t, 1] = float(l1)
                dP[t, 1] = p.grad.numpy()
            st = m.lru.state_dict()
            out = dict(case)
            if tag == "k10":   # the 51000-slot pool is regenerated from the seed by the tests
                touched = sorted(set(v for _, v in st))
                out.pop("queue0")
                out["touched"] = np.asarray(touched, dtype=np.int64)
                out["queue_touched"] = m.queue[:, touched].numpy()
            else:
                out["queue_final"] = m.queue.numpy()
            out.update(loss=loss, dP=dP, lru_keys=np.asarray([k for k, _ in st], dtype=np.int64),
                       lru_slots=np.asarray([v for _, v in st], dtype=np.int64),
                       qp=np.asarray([m.queue_position_dict[i] for i in range(Q)], dtype=np.int8),
                       meta=np.asarray([Q, D, B, T, n_id, m.hard_neg, seed], dtype=np.int64),
                       hyper=np.asarray([32.0, margin], dtype=np.float64))
            np.savez_compressed(os.path.join(HERE, "head_%s_%s.npz" % (loss_type, tag)), **out)
            assert np.isfinite(loss).all(), (loss_type, tag, loss)


# ------------------------------------------------------------------------------------------------
def step_vectors():
    """One full training step (forward, backward, SGD-nesterov step) of the reference FFC on CPU in
    FLOAT64 (so the vectors are the reference's arithmetic without fp32 round-off: deep train-mode BN
    stacks at batch 8 amplify fp32 noise to ~1 % in some weight gradients) for MobileFaceNet and for
    a 4-block iResNet built from the reference's own classes.  Inputs are regenerated from the seed
    by common.step_inputs(), not stored."""
    for tag, net, D, Q, B in (("mobile", "mobile", 32, 64, 8), ("irtiny", "irtiny", 32, 64, 8)):
        seed = {"mobile": 101, "irtiny": 202}[tag]
        rng = np.random.default_rng(seed)
        if net == "mobile":
            sd0, _ = bb.make_backbone("mobile", D)
        else:
            sd0, _ = bb.make_backbone("irtiny", D, layers=(1, 1, 1, 1))
        sd = common.fill_state(sd0, seed)
        m = ref_ffc.FFC("mobile", D, Q, 32.0, "Arc", 0.5, 0.99)
        if net == "irtiny":
            m.probe_net = ref_iresnet.IResNet(ref_iresnet.IBasicBlock, [1, 1, 1, 1], feat_dim=D, fp16=False)  # fp16 only adds a .float() cast (resnet_arcface.py:150)
            m.gallery_net = ref_iresnet.IResNet(ref_iresnet.IBasicBlock, [1, 1, 1, 1], feat_dim=D, fp16=False)  # fp16 only adds a .float() cast (resnet_arcface.py:150)
        missing = m.probe_net.load_state_dict(sd, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        m.gallery_net.load_state_dict(sd, strict=True)
        for pp, pg in zip(m.probe_net.parameters(), m.gallery_net.parameters()):
            pg.requires_grad = False
        if net == "irtiny":
            m.probe_net.features.weight.requires_grad = False
        inp = common.step_inputs(seed, Q, D, B)
        m.double()
        m.queue = torch.from_numpy(inp["queue0"]).double()
        xu8, yu8, xl, yl = inp["xu8"], inp["yu8"], inp["xl"], inp["yl"]
        # warm the pool so the step sees hits, ones_idx and positives (two committing passes)
        warm = inp["warm"].astype(np.float64)
        with torch.no_grad():
            for w in range(2):
                m.probe_net, keep_p = _Pass(), m.probe_net
                m.gallery_net, keep_g = _Pass(), m.gallery_net
                m.forward_impl(torch.from_numpy(warm[w]), torch.from_numpy(warm[1 - w]), torch.from_numpy(xl),
                               torch.from_numpy(yl))
                m.probe_net, m.gallery_net = keep_p, keep_g
        queue_warm = m.queue.numpy().copy()
        lru_warm = m.lru.state_dict()
        qp_warm = np.asarray([m.queue_position_dict[i] for i in range(Q)], dtype=np.int8)

        params = [p for p in m.parameters() if p.requires_grad]
        opt = torch.optim.SGD(params, 0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
        x, y = common.images_from_u8(xu8).double(), common.images_from_u8(yu8).double()
        emb_p = {}
        def _grab(mod, i, o):
            emb_p[len(emb_p)] = o.detach().numpy().copy()

        hooks = [m.probe_net.register_forward_hook(_grab)]
        opt.zero_grad()
        loss = m(x, y, torch.from_numpy(xl), torch.from_numpy(yl))
        loss.backward()
        for h in hooks:
            h.remove()
        names = [n for n, p in m.probe_net.named_parameters() if p.requires_grad]
        grads = {n: p.grad.detach().numpy().copy() for n, p in m.probe_net.named_parameters() if p.requires_grad}
        opt.step()
        after = {n: p.detach().numpy().copy() for n, p in m.probe_net.named_parameters()}
        gal_after = {n: p.detach().numpy().copy() for n, p in m.gallery_net.named_parameters()}
        bufs = {n: b.detach().numpy().copy() for n, b in m.probe_net.named_buffers()}
        gbufs = {n: b.detach().numpy().copy() for n, b in m.gallery_net.named_buffers()}

        def pick(d, keys):
            return {k: d[k] for k in keys}

        if net == "mobile":
            keep = ["conv1.conv.weight", "conv1.bn.weight", "conv1.prelu.weight", "dw_conv1.conv.weight",
                    "blocks.0.conv.0.weight", "blocks.0.conv.3.weight", "blocks.0.conv.6.weight",
                    "blocks.7.conv.4.bias", "linear7.conv.weight", "linear1.conv.weight", "linear1.bn.bias"]
            bkeep = ["conv1.bn.running_mean", "conv1.bn.running_var", "linear1.bn.running_var"]
        else:
            keep = ["conv1.weight", "bn1.weight", "prelu.weight", "layer1.0.conv1.weight", "layer1.0.conv2.weight",
                    "layer1.0.downsample.0.weight", "layer1.0.bn3.bias", "layer2.0.prelu.weight",
                    "layer3.0.conv2.weight", "layer4.0.downsample.1.weight", "bn2.bias", "fc.bias", "features.bias"]
            bkeep = ["bn1.running_mean", "bn1.running_var", "layer4.0.bn3.running_var", "features.running_mean"]
        out = dict(queue_warm=queue_warm, qp_warm=qp_warm,
                   lru_warm_keys=np.asarray([k for k, _ in lru_warm], dtype=np.int64),
                   lru_warm_slots=np.asarray([v for _, v in lru_warm], dtype=np.int64),
                   loss=np.asarray(float(loss)), emb_probe_x=emb_p[0], emb_probe_y=emb_p[1],
                   queue_final=m.queue.numpy(),
                   qp_final=np.asarray([m.queue_position_dict[i] for i in range(Q)], dtype=np.int8),
                   lru_final_keys=np.asarray([k for k, _ in m.lru.state_dict()], dtype=np.int64),
                   lru_final_slots=np.asarray([v for _, v in m.lru.state_dict()], dtype=np.int64),
                   grad_norms=np.asarray([np.linalg.norm(grads[n]) for n in names], dtype=np.float64),
                   meta=np.asarray([Q, D, B, seed], dtype=np.int64))
        out["grad_names"] = np.asarray(names)
        def sample(a):          # large tensors are kept as a strided sample of the flattened array
            flat = a.reshape(-1)
            return flat[::max(1, flat.size // 4096)].copy()

        for k in keep:
            out["grad/" + k] = sample(grads[k])
            out["after/" + k] = sample(after[k])
            out["gallery_after/" + k] = sample(gal_after[k])
        if net == "irtiny":
            out["grad/fc.weight"] = sample(grads["fc.weight"])
        for k in bkeep:
            out["buf/" + k] = bufs[k]
            out["gallery_buf/" + k] = gbufs[k]
        np.savez_compressed(os.path.join(HERE, "step_%s.npz" % tag), **out)
        print(tag, "loss", float(loss))


SCHED_CONFIGS = [
    dict(optim="SGD", scheduler="multistep", LR=0.1, momentum=0.9, decay=1e-4, nesterov=True, warmup=0, epochs=1,
         milestones=[8, 14, 17], gammas=[0.1, 0.1, 0.1]),                                   # config/optim_config
    dict(optim="SGD", scheduler="multistep", LR=0.1, momentum=0.9, decay=1e-4, nesterov=True, warmup=2, epochs=20,
         milestones=[8, 14, 17], gammas=[0.1, 0.5, 0.1]),
    dict(optim="SGD", scheduler="cos", LR=0.05, momentum=0.9, decay=1e-4, nesterov=True, warmup=3, epochs=18, eta_min=1e-5),
    dict(optim="SGD", scheduler="exponential", LR=0.2, momentum=0.9, decay=1e-4, nesterov=True, warmup=1, epochs=12, gamma=0.9),
    dict(optim="SGD", scheduler="linear", LR=0.1, momentum=0.9, decay=1e-4, nesterov=True, warmup=2, epochs=25, LR_min=1e-4),
]


def scheduler_vectors():
    """lr(epoch, iter) of the reference's four warm-up schedules (optim/optimizer.py:47-128) through its own factory
    (:142-168) on a grid of the two update calls main.py makes: update(None, it / len) per iteration (main.py:39-40)
    and update(epoch, 0.0) per epoch (main.py:138-139)."""
    import contextlib
    import io
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_optimizer", "/root/reference/optim/optimizer.py")
    ref_opt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_opt)
    out = []
    for cfg in SCHED_CONFIGS:
        p = [torch.nn.Parameter(torch.zeros(2))]
        with contextlib.redirect_stdout(io.StringIO()):            # CosineAnnealingLR.get_lr prints
            opt, sch = ref_opt.get_optim_scheduler(p, cfg)
            rows = []
            for epoch in range(0, cfg["epochs"] + cfg["warmup"] + 3):
                sch.update(epoch, 0.0)
                rows.append([epoch, 0.0, opt.param_groups[0]["lr"]])
                for it in (0.25, 0.5, 0.875):
                    sch.update(None, it)
                    rows.append([epoch, it, opt.param_groups[0]["lr"]])
        out.append(dict(config=cfg, rows=rows))
    with open(os.path.join(HERE, "scheduler_lrs.json"), "w") as f:
        json.dump(out, f)
    print("scheduler_lrs.json: %d configs, %d points" % (len(out), sum(len(o["rows"]) for o in out)))


def _sample(a):          # large tensors are kept as a strided sample of the flattened array (as in step_vectors)
    flat = np.asarray(a).reshape(-1)
    return flat[::max(1, flat.size // 4096)]


def resnet_std_vectors():
    """Forward + backward of the reference's torchvision-style ResNet (model/resnet_std.py) in float64 on CPU: one
    Bottleneck per stage (the class the reference builds r50 from, :242-251), feat_dim 32, eight 224 x 224 images, seeded
    weights (common.fill_state over the reference's own state dict names) -> embeddings, the gradient norm of every
    parameter and samples of the gradient tensors for the loss sum(emb * c), c a seeded constant."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_resnet_std", "/root/reference/model/resnet_std.py")
    ref_std = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_std)
    net = ref_std.ResNet(ref_std.Bottleneck, [1, 1, 1, 1], feat_dim=32).double()
    sd = common.fill_state({k: (v.detach().float() if v.is_floating_point() else v.detach().clone())
                            for k, v in net.state_dict().items()}, 41)
    net.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
    net.train()
    rng = np.random.default_rng(41)
    x = common.images_from_u8(common.synth_images_u8(rng, 8, hw=224)).double()
    c = torch.from_numpy(rng.standard_normal((8, 32)))
    emb = net(x)
    (emb * c).sum().backward()
    out = dict(meta=np.asarray([32, 8, 41, 224]), emb=emb.detach().numpy(), c=c.numpy(),
               grad_names=np.asarray([n for n, _ in net.named_parameters()]),
               grad_norms=np.asarray([float(p.grad.norm()) for _, p in net.named_parameters()]))
    for n, p in net.named_parameters():
        out["grad/" + n] = _sample(p.grad.detach().numpy()).astype(np.float32)
    for n, b in net.named_buffers():
        if n.endswith("running_mean") and ("layer4" in n or n in ("bn1.running_mean", "features.running_mean")):
            out["buf/" + n] = b.detach().numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "backbone_rstd.npz"), **out)
    print("backbone_rstd.npz: emb", out["emb"].shape, "%d gradient tensors" % len(out["grad_names"]))


if __name__ == "__main__":
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "schedulers":
        scheduler_vectors()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "resnet_std":
        resnet_std_vectors()
        sys.exit(0)
    resnet_std_vectors()
    scheduler_vectors()
    lru_traces()
    head_vectors()
    step_vectors()
    print("golden vectors written to", HERE)
