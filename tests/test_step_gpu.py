"""Whole training step on the GPU (native iResNet executor + fused head + fused SGD/EMA, through the
drop-in FFC / get_optim_scheduler surface) against the reference's golden step vectors (float64
arithmetic of the reference itself) and against the float64 oracle.

Tolerances: the backbone computes with bf16 operands / fp32 accumulation, so
embeddings are compared by cosine (>= 0.999), the loss to rtol 2e-2, gradient tensors by relative
L2 error, LRU / queue_position state exactly."""
import os

import numpy as np
import pytest
import torch

from oracle import backbones_ref as bb
from tests.golden import common
from tests.test_oracle_golden import G, build_oracle_from_step, sample

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def min_cos(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float(torch.nn.functional.cosine_similarity(a, b, dim=1).min())


def build_ffc(z, net_type, precise_head=True):
    from vlsfr_amd.ffc import FFC
    Q, D, B, seed = [int(v) for v in z["meta"]]
    m = FFC(net_type, D, Q, 32.0, "Arc", 0.5, 0.99, precise_head=precise_head)
    sd0, _ = bb.make_backbone(net_type, D, layers=(1, 1, 1, 1) if net_type == "irtiny" else None)
    sd = common.fill_state(sd0, seed)
    m.probe_net.load_state_dict(sd, strict=True)
    m.gallery_net.load_state_dict(sd, strict=True)
    m = m.cuda()
    m.queue.copy_(torch.from_numpy(z["queue_warm"]).float())
    m.lru.restore(list(zip(z["lru_warm_keys"].tolist(), z["lru_warm_slots"].tolist())))
    m._state().qp[:] = z["qp_warm"].astype(np.uint8)
    inp = common.step_inputs(seed, Q, D, B)
    x, y = common.images_from_u8(inp["xu8"]).cuda(), common.images_from_u8(inp["yu8"]).cuda()
    return m, x, y, torch.from_numpy(inp["xl"]), torch.from_numpy(inp["yl"])


@pytest.mark.parametrize("tag", ["irtiny", "mobile"])
def test_step_matches_reference_golden(tag):
    from vlsfr_amd.optim import get_optim_scheduler
    z = np.load(os.path.join(G, "step_%s.npz" % tag))
    m, x, y, xl, yl = build_ffc(z, tag)
    cfg = dict(optim="SGD", scheduler="multistep", LR=0.1, momentum=0.9, decay=1e-4, nesterov=True, warmup=0, epochs=1,
               milestones=[8, 14, 17], gammas=[0.1, 0.1, 0.1])
    opt, sched = get_optim_scheduler([p for p in m.parameters() if p.requires_grad], cfg)
    sched.update(0, 0.0)
    embs = []
    hook = m.probe_net.register_forward_hook(lambda mod, i, o: embs.append(o.detach().cpu()))
    opt.zero_grad()
    loss = m(x, y, xl, yl)
    loss.backward()
    hook.remove()
    torch.cuda.synchronize()
    # embeddings, loss.  MobileFaceNet stacks 50 bf16 conv + train-mode BN layers and ends in two
    # BatchNorms over only 8 samples (1x1 maps), so its bf16 round-off is amplified more than the
    # 4-block iResNet's: the bounds are looser there (scripts/bf16_sensitivity.py calibrates this on
    # the CPU: fp32-vs-fp64 deviation x 2^15 predicts ~9 % for MobileFaceNet, ~3 % for the iResNet).
    tol = dict(irtiny=dict(cos=0.999, loss=2e-2, gnorm=8e-2, gl2=0.1), mobile=dict(cos=0.995, loss=3e-2, gnorm=0.25, gl2=0.45))[tag]
    assert min_cos(embs[0], z["emb_probe_x"]) >= tol["cos"]
    assert min_cos(embs[1], z["emb_probe_y"]) >= tol["cos"]
    np.testing.assert_allclose(float(loss.detach()), float(z["loss"]), rtol=tol["loss"])
    # gradients.  The calibrated band (ADVICE r01 / VERDICT r02): what bf16 STORAGE alone does to this step — the float64
    # oracle with the device's rounding points against the reference's plain float64 vectors.  The device may deviate from the
    # reference by at most twice that, per tensor (or twice the band over all sampled tensors, where a tensor's own is tiny),
    # and never by more than the absolute cap in `tol`.
    from tests.test_oracle_golden import build_oracle_from_step
    oe, xe, ye, xle, yle = build_oracle_from_step(z, tag, emulate_bf16=True)
    oe.forward(xe, ye, xle, yle).backward()
    band = {}
    for key in z.files:
        if key.startswith("grad/") and np.abs(z[key]).max() >= 1e-6:
            band[key] = rel_l2(sample(oe.probe[key[5:]].grad.numpy()), z[key])
    num = sum(float(((sample(oe.probe[k[5:]].grad.numpy()) - z[k]) ** 2).sum()) for k in band)
    band_all = float(np.sqrt(num / sum(float((z[k] ** 2).sum()) for k in band)))
    names = [str(n) for n in z["grad_names"]]
    pn = dict(m.probe_net.named_parameters())
    gn = np.asarray([float(pn[n].grad.norm()) for n in names])
    big = z["grad_norms"] > 1e-3 * z["grad_norms"].max()
    np.testing.assert_allclose(gn[big], z["grad_norms"][big], rtol=min(tol["gnorm"], 2 * band_all + 2e-2))
    worst = (None, 0.0, 0.0)
    for key in band:
        got = sample(pn[key[5:]].grad.detach().cpu().numpy())
        err, lim = rel_l2(got, z[key]), min(tol["gl2"], 2.0 * max(band[key], band_all, 1e-2))
        if err / lim > worst[1] / max(worst[2], 1e-30):
            worst = (key, err, lim)
        assert err <= lim, (key, err, band[key], band_all)
    print("%s: bf16-storage band over the sampled gradients %.3f; tightest tensor %s: device %.3f, bound %.3f" % ((tag, band_all) + worst))
    # optimizer step + EMA
    opt.step()
    torch.cuda.synchronize()
    gp = dict(m.gallery_net.named_parameters())
    for key in z.files:
        if key.startswith("after/"):
            got = sample(pn[key[6:]].detach().cpu().numpy())
            assert rel_l2(got, z[key]) < tol["gl2"], (key, rel_l2(got, z[key]))   # lr * gradient dominates: same bound as the gradients
        elif key.startswith("gallery_after/"):
            np.testing.assert_allclose(sample(gp[key[14:]].detach().cpu().numpy()), z[key], rtol=1e-5, atol=1e-6)
        elif key.startswith("buf/"):
            got = dict(m.probe_net.named_buffers())[key[4:]].cpu().numpy()
            np.testing.assert_allclose(got, z[key], rtol=3e-2, atol=3e-3)
    # pool and allocator state
    assert [k for k, _ in m.lru.state_dict()] == z["lru_final_keys"].tolist()
    assert [v for _, v in m.lru.state_dict()] == z["lru_final_slots"].tolist()
    assert m.queue_position_dict.values() == z["qp_final"].astype(int).tolist()
    qf, qw = m.queue.cpu().numpy(), z["queue_final"]
    changed = np.abs(qw - z["queue_warm"]).max(axis=2) > 0
    assert min_cos(qf[changed], qw[changed]) >= tol["cos"]           # rows written by the gallery net
    np.testing.assert_array_equal(qf[~changed], qw[~changed].astype(np.float32))


@pytest.mark.parametrize("tag", ["irtiny", "mobile"])
def test_streams_do_not_change_the_step(tag):
    """The three-stream schedule (gallery pass, second backward pass and the rollback head beside the
    main stream) against the single-stream order on the same inputs: loss, pool, LRU state and every
    gradient tensor (fp32 atomics reorder sums, hence a tolerance instead of bit equality)."""
    z = np.load(os.path.join(G, "step_%s.npz" % tag))
    outs = []
    for concurrent, chains in ((False, 2), (False, 2), (True, 2), (True, 4)):
        m, x, y, xl, yl = build_ffc(z, tag)
        m.concurrent_streams = concurrent
        m.forward_chains = chains
        m.probe_net.concurrent_backward = concurrent
        loss = m(x, y, xl, yl)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((float(loss.detach()), m.queue.clone(), m.lru.state_dict(), m._state().qp.copy(),
                     {k: p.grad.clone() for k, p in m.probe_net.named_parameters() if p.grad is not None},
                     {n + k: b.clone() for n, net in (("p.", m.probe_net), ("g.", m.gallery_net)) for k, b in net.named_buffers()
                      if k.endswith("running_mean") or k.endswith("running_var")}))
    # four chains (FFC.embed_both: both passes' backbones side by side, running statistics merged afterwards): the same
    # bounds as the two-chain schedule, and the running statistics after the two updates of the step
    (l0, q0, s0, qp0, g0, r0), (l4, q4, s4, qp4, g4, r4) = outs[0], outs[3]
    assert r0.keys() == r4.keys() and len(r0) >= 20
    for k in r0:
        scale = float(r0[k].abs().max()) + 1e-6
        assert float((r0[k] - r4[k]).abs().max()) <= (2e-2 if tag == "mobile" else 2e-3) * scale, k
    outs = [o[:5] for o in outs]
    for (l0, q0, s0, qp0, g0), (l0b, q0b, _, _, g0b), (l1, q1, s1, qp1, g1) in ((outs[0], outs[1], outs[2]), (outs[0], outs[1], outs[3])):
        _compare_schedules(tag, (l0, q0, s0, qp0, g0), (l0b, q0b, g0b), (l1, q1, s1, qp1, g1))


def _compare_schedules(tag, a, b, c):
    (l0, q0, s0, qp0, g0), (l0b, q0b, g0b), (l1, q1, s1, qp1, g1) = a, b, c
    # Run-to-run noise of the single-stream order itself: the BN statistics are summed with fp32 atomics in
    # arrival order and the train-mode BN stack at batch 8 amplifies that round-off (MobileFaceNet's loss moves
    # by up to 5e-3 between two identical single-stream runs, scripts/noise_test.py).  The stream schedule has to
    # stay inside that band; the integer state must be identical.
    cat = lambda g: np.concatenate([g[k].float().cpu().numpy().ravel() for k in sorted(g)])
    noise_l = abs(l0 - l0b) / abs(l0)
    noise_g = rel_l2(cat(g0b), cat(g0))
    noise_q = float((q0 - q0b).abs().max())
    d_l, d_g, d_q = abs(l0 - l1) / abs(l0), rel_l2(cat(g1), cat(g0)), float((q0 - q1).abs().max())
    print("single-stream run-to-run: loss %.2e, gradients rel-L2 %.2e, pool rows %.2e" % (noise_l, noise_g, noise_q))
    print("streams vs single stream: loss %.2e, gradients rel-L2 %.2e, pool rows %.2e" % (d_l, d_g, d_q))
    loose = tag == "mobile"
    assert d_l <= max(4 * noise_l, 1.5e-2 if loose else 2e-3)
    assert d_q <= max(4 * noise_q, 0.1 if loose else 5e-3)
    assert d_g <= max(4 * noise_g, 0.3 if loose else 5e-2)
    assert s0 == s1 and (qp0 == qp1).all()
    assert g0.keys() == g1.keys()


@pytest.mark.parametrize("tag", ["mobile", "irtiny"])
def test_graph_replay_equals_plain_launches(tag):
    """NativeBackbone.use_graphs: the forward / backward executor calls of a step replayed from HIP graphs (captured on the
    third call with the same buffers) against plain launches.  Five steps on the same inputs with a zero learning rate (the
    weights stay put, so every step computes the same thing and the comparison is not a chaotic training trajectory): the
    loss of every step, the gradients of the last one and the BatchNorm running statistics agree within the run-to-run
    noise of the atomically summed statistics; the allocator state is identical."""
    from vlsfr_amd.optim.fused import FusedSGD
    z = np.load(os.path.join(G, "step_%s.npz" % tag))
    traj = []
    for graphs in (False, True):
        m, x, y, xl, yl = build_ffc(z, tag)
        m.probe_net.use_graphs = m.gallery_net.use_graphs = graphs
        opt = FusedSGD([p for p in m.parameters() if p.requires_grad], 0.0, momentum=0.0, weight_decay=0.0, nesterov=False)
        losses = []
        gen = torch.Generator(device="cuda").manual_seed(5)
        for _ in range(5):
            opt.zero_grad()
            # fresh pixel noise per step: the very same image twice meets its own gallery row at cos = 1, where the Arc margin
            # of the reference has no finite gradient
            xs = x + 0.1 * torch.randn(x.shape, device="cuda", generator=gen)
            ys = y + 0.1 * torch.randn(y.shape, device="cuda", generator=gen)
            loss = m(xs, ys, xl, yl)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        if graphs:
            captured = [k for k, st in m.probe_net._graphs.items() if st["graph"] is not None]
            assert any(k[0] == "fwd" for k in captured) and any(k[0] == "bwd" for k in captured), captured
        traj.append((losses, {k: p.grad.detach().clone() for k, p in m.probe_net.named_parameters() if p.grad is not None},
                     {k: b.detach().clone() for k, b in m.probe_net.named_buffers() if k.endswith("running_var")},
                     m.lru.state_dict(), m._state().qp.copy()))
    (l0, g0, r0, s0, q0), (l1, g1, r1, s1, q1) = traj
    print("losses plain %s\n       graph %s" % (l0, l1))
    loose = tag == "mobile"
    assert np.isfinite(l1).all()
    np.testing.assert_allclose(l1, l0, rtol=2e-2 if loose else 3e-3)
    cat = lambda d: np.concatenate([d[k].float().cpu().numpy().ravel() for k in sorted(d)])
    assert rel_l2(cat(g1), cat(g0)) <= (0.3 if loose else 5e-2)          # the bounds of test_streams_do_not_change_the_step
    assert rel_l2(cat(r1), cat(r0)) <= 1e-3
    assert s0 == s1 and (q0 == q1).all()


@pytest.mark.parametrize("tag", ["irtiny", "mobile"])
def test_graph_replay_returns_the_plain_embedding(tag):
    """The embedding a captured forward pass hands out (a copy KERNEL out of the executor's buffer, no runtime copy node) against
    plain launches on the same input: equal up to what two plain launches differ by themselves (the float64 statistics are summed
    atomically: their order is the only noise), on every replay — a stale or half-written embedding would be off by orders more."""
    z = np.load(os.path.join(G, "step_%s.npz" % tag))
    m, x, y, xl, yl = build_ffc(z, tag)
    net = m.gallery_net
    with torch.no_grad():
        plain = [net(x).clone() for _ in range(3)]
        noise = max(float((plain[0] - p).abs().max()) for p in plain[1:])
        net.use_graphs = True
        got = [net(x).clone() for _ in range(8)]          # 2 eager warm-ups, capture, 5 replays
        torch.cuda.synchronize()
        assert any(k[0] == "fwd" and st["graph"] is not None for k, st in net._graphs.items())
        net.use_graphs = False
    for g in got:
        assert float((g - plain[0]).abs().max()) <= max(4 * noise, 1e-6), (float((g - plain[0]).abs().max()), noise)
    assert float((net(x) - plain[0]).abs().max()) <= max(4 * noise, 1e-6)


def test_ir18_two_steps_vs_oracle():
    """A deeper net (ir18), two consecutive steps, against the float64 oracle: loss trajectory,
    LRU / queue_position state, embedding cosine."""
    from vlsfr_amd.ffc import FFC
    from vlsfr_amd.optim.fused import FusedSGD
    from oracle import ffc_ref
    torch.manual_seed(0)
    Q, D, B = 96, 64, 8
    o = ffc_ref.FFCRef("ir18", D, Q, 32.0, "AM", 0.4, 0.99, dtype=torch.float64)
    sd = common.fill_state({k: v.detach() for k, v in o.probe.items()}, 77)
    o.probe = {k: (v.double().requires_grad_(bb.trainable(k)) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    o.gallery = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    m = FFC("ir18", D, Q, 32.0, "AM", 0.4, 0.99, precise_head=True)
    m.probe_net.load_state_dict(sd)
    m.gallery_net.load_state_dict(sd)
    m = m.cuda()
    m.queue.copy_(o.queue.float())
    opt = FusedSGD([p for p in m.parameters() if p.requires_grad], 0.05, momentum=0.9, weight_decay=1e-4, nesterov=True)
    rng = np.random.default_rng(9)
    bufs = None
    for step in range(2):
        xu8, yu8 = common.synth_images_u8(rng, B), common.synth_images_u8(rng, B)
        ids = rng.choice(50, size=B // 2, replace=False)
        xl = torch.from_numpy(np.concatenate([ids, rng.integers(0, 50, B // 2)]).astype(np.int64))
        yl = torch.from_numpy(np.concatenate([ids, rng.integers(0, 50, B // 2)]).astype(np.int64))
        x, y = common.images_from_u8(xu8), common.images_from_u8(yu8)
        params = o.parameters()
        for p in params:
            p.grad = None
        lo = o.forward(x.double(), y.double(), xl, yl)
        lo.backward()
        bufs = ffc_ref.sgd_nesterov_step_ref(params, [p.grad for p in params], bufs or [None] * len(params), 0.05)
        opt.zero_grad()
        lg = m(x.cuda(), y.cuda(), xl, yl)
        lg.backward()
        opt.step()
        # step 0 starts from identical weights; step 1 inherits the (lr-amplified) bf16 gradient noise of step 0
        np.testing.assert_allclose(float(lg.detach()), float(lo.detach()), rtol=3e-2 if step == 0 else 1e-1)
        assert m.lru.state_dict() == o.lru.state_dict()
        assert m.queue_position_dict.values() == o.qp
    w_o = o.probe["layer2.0.conv1.weight"].detach().numpy()
    w_g = m.probe_net.layer2[0].conv1.weight.detach().cpu().numpy()
    # lr * gradient dominates the two-step weight change, and bf16 gradients of this deep, batch-8
    # train-mode-BN stack deviate 3-10 % from float64 (scripts/diag_grad_profile.py shows the error
    # growing smoothly from 1 % at the head to 10 % at the stem, no jump at any layer type)
    assert rel_l2(w_g, w_o) < 0.25


def test_main_train_loop(tmp_path):
    """The training driver (main.py mirror): a few iterations on synthetic faces, checkpoint format of
    the reference (main.py:85)."""
    from vlsfr_amd.main import parse_args, train
    conf = parse_args(["--net_type", "irtiny", "--feat_dim", "32", "--queue_size", "64", "--batch_size", "8",
                       "--print_freq", "3", "--iters_per_epoch", "6", "--num_class", "500", "--saved_dir", str(tmp_path)])
    logs = []
    net, loss = train(conf, log=logs.append)
    assert np.isfinite(float(loss.detach())) and len(logs) == 2
    ck = torch.load(os.path.join(str(tmp_path), "2.pt"), weights_only=False)
    assert set(ck) == {"state_dict", "lru", "fc", "qp", "resume"}      # the reference's four keys + the resume extras
    assert ck["fc"].shape == (2, 64, 32) and len(ck["qp"]) == 64 and len(ck["lru"]) == len(net.lru.state_dict())
    assert "layer1.0.conv1.weight" in ck["state_dict"]
    assert int(ck["state_dict"]["bn1.num_batches_tracked"]) == 12      # two probe forwards per iteration


def test_resume_continues_the_run(tmp_path):
    """Checkpoint -> `--resume`: a run interrupted after 2 of 4 iterations and resumed reaches the state
    of the uninterrupted run -- LRU order and queue positions exactly, pool rows and loss within the run-to-run
    noise of the bf16 / atomic-order arithmetic."""
    from vlsfr_amd.main import parse_args, train
    base = ["--net_type", "irtiny", "--feat_dim", "32", "--queue_size", "64", "--batch_size", "8", "--print_freq", "2",
            "--iters_per_epoch", "4", "--num_class", "500"]
    d1, d2 = tmp_path / "a", tmp_path / "b"
    net_a, loss_a = train(parse_args(base + ["--saved_dir", str(d1)]), log=lambda *_: None)
    net_b, loss_b = train(parse_args(base + ["--saved_dir", str(d2), "--resume", str(d1 / "1.pt")]), log=lambda *_: None)
    assert net_a.lru.state_dict() == net_b.lru.state_dict()
    assert net_a.queue_position_dict.to_dict() == net_b.queue_position_dict.to_dict()
    la, lb = float(loss_a.detach()), float(loss_b.detach())
    assert abs(la - lb) <= 2e-2 * abs(la)
    assert float((net_a.queue - net_b.queue).abs().max()) < 0.05
    wa = net_a.probe_net.state_dict()["layer1.0.conv1.weight"].float().cpu().numpy()
    wb = net_b.probe_net.state_dict()["layer1.0.conv1.weight"].float().cpu().numpy()
    assert rel_l2(wb, wa) < 2e-2


def _emulated_oracle(net_type, D, Q, B, loss_type, margin, seed, layers, dtype):
    """The oracle with bf16-storage emulation (oracle/backbones_ref.py) on seeded weights, pool and batch."""
    from oracle import ffc_ref
    o = ffc_ref.FFCRef(net_type, D, Q, 32.0, loss_type, margin, 0.99, layers=layers, dtype=dtype, emulate_bf16=True)
    sd = common.fill_state({k: v.detach() for k, v in o.probe.items()}, seed)
    o.probe = {k: (v.to(dtype).requires_grad_(bb.trainable(k, net_type)) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    o.gallery = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    rng = np.random.default_rng(seed)
    o.queue = torch.from_numpy(common.unit_rows(rng, 2, Q, D)).to(dtype)
    n_id = max(Q // 2, 16)
    ids = rng.choice(n_id, size=B // 2, replace=False)
    xl = torch.from_numpy(np.concatenate([ids, rng.integers(0, n_id, B - B // 2)]).astype(np.int64))
    yl = torch.from_numpy(np.concatenate([ids, rng.integers(0, n_id, B - B // 2)]).astype(np.int64))
    hw = 64 if net_type == "rtiny" else 112
    x = common.images_from_u8(common.synth_images_u8(rng, B, hw=hw))
    y = common.images_from_u8(common.synth_images_u8(rng, B, hw=hw))
    return o, sd, x, y, xl, yl


def _emulated_pair(net_type, D, Q, B, loss_type, margin, seed, layers=None, head_dtype=None):
    """The product FFC on the GPU and the float64 emulating oracle, same seeded weights, pool and batch.  head_dtype
    None: the head in precise mode (split-bf16 products: fp32-class), so that the comparison isolates the backbone;
    "bf16" / "fp8": the shadow sweeps the benchmarks run (csrc/head16.hip / head8.hip; D = 512)."""
    from vlsfr_amd.ffc import FFC
    o, sd, x, y, xl, yl = _emulated_oracle(net_type, D, Q, B, loss_type, margin, seed, layers, torch.float64)
    m = FFC(net_type, D, Q, 32.0, loss_type, margin, 0.99, precise_head=head_dtype is None)
    if head_dtype is not None:
        m.__dict__['head_dtype'] = head_dtype
    m.probe_net.load_state_dict(sd)
    m.gallery_net.load_state_dict(sd)
    m = m.cuda()
    m.queue.copy_(o.queue.float())
    return m, o, x, y, xl, yl


# (net, feat_dim, pool slots, batch_size): the two backbones the bench runs (ir50 = BASELINE configs[1], ir100 = the
# metric) at the batch the float64 oracle finishes in seconds, the 4-block iResNet and MobileFaceNet at batch 32.
# "mobile512-fp8" is BASELINE configs[4] (C5) in one piece: MobileFaceNet, D = 512, the e4m3 class matmul.
EMU_CASES = [("irtiny", 64, 512, 32, (1, 1, 1, 1), None), ("mobile", 128, 1000, 32, None, None), ("ir50", 512, 2048, 8, None, None),
             ("ir100", 512, 2048, 8, None, None), ("rtiny", 64, 512, 16, None, None),
             ("mobile", 512, 4096, 32, None, "fp8"), ("mobile", 512, 4096, 32, None, "bf16")]
EMU_IDS = [c[0] + ("%d-%s" % (c[1], c[5]) if c[5] else "") for c in EMU_CASES]
# What the reduced-precision class matmul may add to the backbone's band (bf16 loss 2e-2, fp8 loss 5e-2 and
# cosine >= 0.99; tests/test_head_gpu.py measures the heads alone: loss 3e-3 / 5e-3, dL/dp rel-L2 <= 8 % for fp8).
HEAD_EXTRA = {None: dict(loss=0.0, g=0.0), "bf16": dict(loss=3e-3, g=2e-2), "fp8": dict(loss=5e-3, g=8e-2)}
# Absolute caps on top of the self-calibrated band below: loss, embedding cosine (>= 0.999 for bf16), relative
# L2 of the whole gradient.  (The per-tensor bound is purely band-relative: a tensor's own band can be large where its
# exact gradient is tiny.)
EMU_CAP = {"irtiny": dict(loss=1e-3, cos=0.9995, g_all=5e-2), "mobile": dict(loss=2e-3, cos=0.999, g_all=0.35),
           "ir50": dict(loss=5e-3, cos=0.999, g_all=0.2), "ir100": dict(loss=5e-3, cos=0.999, g_all=0.25),
           "rtiny": dict(loss=2e-3, cos=0.999, g_all=0.2)}


def _grad_errors(got, ref):
    """Relative L2 error of the whole gradient and per tensor; tensors whose exact gradient is zero (a bias in front
    of a BatchNorm) are left out of the per-tensor list."""
    num = sum(float(((got[k] - ref[k]) ** 2).sum()) for k in ref)
    den = sum(float((ref[k] ** 2).sum()) for k in ref)
    typical = np.sqrt(den / sum(ref[k].size for k in ref))
    each = {k: rel_l2(got[k], ref[k]) for k in ref if np.sqrt((ref[k] ** 2).mean()) > 1e-2 * typical}
    return float(np.sqrt(num / den)), each


@pytest.mark.parametrize("net_type,D,Q,B,layers,head_dtype", EMU_CASES, ids=EMU_IDS)
def test_step_vs_bf16_emulating_oracle(net_type, D, Q, B, layers, head_dtype):
    """One whole training step (probe / gallery backbones, both DCP passes, backward, SGD-nesterov, EMA) of the
    benchmarked networks against the float64 oracle that rounds to bf16 exactly where the device stores bf16: loss,
    embeddings, EVERY gradient tensor, post-step parameters, gallery EMA, LRU / queue_position state.

    Gradient bound, self-calibrated: the same emulating oracle evaluated in float32 differs from its float64 self
    only by summation rounding, which the bf16 re-rounding of every stored tensor and the train-mode BatchNorm stack
    amplify into a configuration-dependent band (measured, gpurun_out r2e: whole-gradient relative L2 of CPU fp32 vs
    CPU fp64 / of GPU vs CPU fp64 = 1.92e-2 / 1.95e-2 for the 4-block iResNet at batch 32, 0.233 / 0.232 for
    MobileFaceNet at batch 32, 0.096 / 0.097 for ir50 and 0.141 / 0.141 for ir100 at batch 8 — the GPU sits ON the
    band of an fp32 evaluation of the same rounding-point model, per layer too: scripts/diag_emulated_grads.py).
    The GPU's whole gradient must lie within 1.5x that band, every tensor within 2x its own (or the overall) band,
    and everything under the absolute caps of EMU_CAP."""
    from vlsfr_amd.optim.fused import FusedSGD
    from oracle import ffc_ref
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    m, o, x, y, xl, yl = _emulated_pair(net_type, D, Q, B, "Arc", 0.5, 31, layers, head_dtype)
    cap, extra = EMU_CAP[net_type], HEAD_EXTRA[head_dtype]
    embs = []
    hook = m.probe_net.register_forward_hook(lambda mod, i, out: embs.append(out.detach().cpu()))
    opt = FusedSGD([p for p in m.parameters() if p.requires_grad], 0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    opt.zero_grad()
    loss = m(x.cuda(), y.cuda(), xl, yl)
    loss.backward()
    hook.remove()
    torch.cuda.synchronize()
    with torch.no_grad():
        emb_x = o.fwd({k: v.clone() for k, v in o.probe.items()}, x.double())      # probe(x) before any update
    want = o.forward(x.double(), y.double(), xl, yl)
    want.backward()
    o32, _, _, _, _, _ = _emulated_oracle(net_type, D, Q, B, "Arc", 0.5, 31, layers, torch.float32)
    w32 = o32.forward(x, y, xl, yl)
    w32.backward()
    ref = {k: v.grad.numpy() for k, v in o.probe.items() if bb.trainable(k, net_type)}
    f32 = {k: v.grad.double().numpy() for k, v in o32.probe.items() if bb.trainable(k, net_type)}
    pn = dict(m.probe_net.named_parameters())
    got = {k: pn[k].grad.detach().double().cpu().numpy() for k in ref}
    noise_all, noise_each = _grad_errors(f32, ref)
    g_all, g_each = _grad_errors(got, ref)
    worst = max(g_each, key=lambda k: g_each[k] / max(noise_each[k], noise_all))
    print("%s: loss gpu %.6f / fp64 %.6f / fp32 %.6f; gradient rel-L2: gpu %.2e (fp32-vs-fp64 band %.2e); worst tensor "
          "%s gpu %.2e band %.2e" % (net_type, float(loss.detach()), float(want.detach()), float(w32.detach()), g_all,
                                     noise_all, worst, g_each[worst], noise_each[worst]))
    np.testing.assert_allclose(float(loss.detach()), float(want.detach()),
                               rtol=extra["loss"] + max(cap["loss"], 3 * abs(float(w32.detach()) - float(want.detach())) / float(want.detach())))
    assert min_cos(embs[0], emb_x) >= cap["cos"]
    assert m.lru.state_dict() == o.lru.state_dict()
    assert m.queue_position_dict.values() == o.qp
    if head_dtype is not None:
        assert m._ensure_head().shadow.t.get(head_dtype) is not None            # the shadow sweep is what ran
    assert g_all <= extra["g"] + min(cap["g_all"], max(1.5 * noise_all, 5e-3)), (g_all, noise_all)
    for k, e in g_each.items():
        assert e <= extra["g"] + 2.0 * max(noise_each[k], noise_all, 2e-3), (k, e, noise_each[k], noise_all)
    # SGD-nesterov step + what the EMA made of the gallery net (the EMA ran inside forward, before the step)
    ps = o.parameters()
    before = {k: v.detach().clone() for k, v in o.probe.items() if bb.trainable(k, net_type)}
    ffc_ref.sgd_nesterov_step_ref(ps, [p.grad for p in ps], [None] * len(ps), 0.1)
    opt.step()
    torch.cuda.synchronize()
    num = den = 0.0
    for k, v in o.probe.items():
        if bb.trainable(k, net_type):
            gk = pn[k].detach().double().cpu().numpy()
            num += float(((gk - v.detach().numpy()) ** 2).sum())
            den += float(((v.detach() - before[k]).numpy() ** 2).sum())
    assert np.sqrt(num / den) <= extra["g"] + min(cap["g_all"], max(1.5 * noise_all, 5e-3))     # error relative to the size of the update
    gp = dict(m.gallery_net.named_parameters())
    for k, v in o.gallery.items():
        if not bb.is_buffer(k):
            np.testing.assert_allclose(gp[k].detach().cpu().numpy(), v.float().numpy(), rtol=1e-5, atol=1e-6)


def test_reference_loop_shape_with_autocast_and_gradscaler():
    """The reference's loop (main.py:64-71,133): `with torch.amp.autocast('cuda')` around the forward,
    `scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()`.  The MI355X path needs neither (bf16
    operands have fp32's exponent range) — leaving them in must be harmless: same loss, same updated parameters (the
    loss scale is a power of two: scaling and unscaling are exact; what remains is the run-to-run noise of the atomic
    summation order, measured here by a second plain run), same pool and allocator state."""
    from vlsfr_amd.optim import get_optim_scheduler
    z = np.load(os.path.join(G, "step_irtiny.npz"))
    cfg = dict(optim="SGD", scheduler="multistep", LR=0.1, momentum=0.9, decay=1e-4, nesterov=True, warmup=0, epochs=1,
               milestones=[8, 14, 17], gammas=[0.1, 0.1, 0.1])
    outs = []
    for mode in ("plain", "plain", "amp"):
        m, x, y, xl, yl = build_ffc(z, "irtiny")
        opt, sched = get_optim_scheduler([p for p in m.parameters() if p.requires_grad], cfg)
        sched.update(None, 0.0)
        opt.zero_grad()
        if mode == "amp":
            scaler = torch.amp.GradScaler("cuda")
            with torch.amp.autocast("cuda"):
                loss = m(x, y, xl, yl)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            assert scaler.get_scale() == 65536.0          # no inf / nan was found: the step was taken
        else:
            loss = m(x, y, xl, yl)
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        assert loss.dtype == torch.float32
        outs.append((float(loss.detach()), np.concatenate([p.detach().float().cpu().numpy().ravel() for p in m.probe_net.parameters()]),
                     m.queue.clone(), m.lru.state_dict(), m._state().qp.copy()))
    (l0, w0, q0, s0, qp0), (l1, w1, q1, _, _), (l2, w2, q2, s2, qp2) = outs
    noise_l, noise_w, noise_q = abs(l0 - l1) / abs(l0), rel_l2(w1, w0), float((q0 - q1).abs().max())
    assert abs(l0 - l2) / abs(l0) <= max(4 * noise_l, 1e-3)
    assert rel_l2(w2, w0) <= max(4 * noise_w, 1e-4)
    assert s0 == s2 and (qp0 == qp2).all()
    assert float((q0 - q2).abs().max()) <= max(4 * noise_q, 1e-2)      # pool rows are unit vectors written by the gallery net


def test_main_trains_from_a_face_store(tmp_path):
    """main.py's loop fed by the reference's two datasets over a FaceStore (data.py), images normalised on the GPU."""
    from vlsfr_amd.data import make_synthetic_store
    from vlsfr_amd.main import parse_args, train
    store, kv = make_synthetic_store(str(tmp_path / "db"), "faces", 40, 3, hw=112, seed=4)
    conf = parse_args(["--net_type", "irtiny", "--feat_dim", "32", "--queue_size", "64", "--batch_size", "8", "--print_freq", "2",
                       "--iters_per_epoch", "3", "--saved_dir", "", "--data_store", store, "--data_kv", kv])
    net, loss = train(conf, log=lambda *_: None)
    assert np.isfinite(float(loss.detach()))
    assert len(net.lru.state_dict()) > 0 and max(k for k, _ in net.lru.state_dict()) < 40      # labels come from the kv file


def test_resnet_std_backbone_matches_reference_golden():
    """The torchvision-style ResNet executor (csrc/resnet.cpp; `--net_type r50` family) on the reference's own float64
    outputs (tests/golden/backbone_rstd.npz: model/resnet_std.py ResNet(Bottleneck, [1,1,1,1]), eight 224 x 224 images):
    embeddings by cosine, gradient norms and sampled gradient tensors within the bf16 band (calibrated like
    test_step_vs_bf16_emulating_oracle: GPU vs the emulating oracle must sit on the fp32-vs-fp64 band)."""
    from tests.test_oracle_golden import build_rstd_oracle
    from vlsfr_amd.model.resnet_std import ResNet
    z = np.load(os.path.join(G, "backbone_rstd.npz"))
    D, B, seed, hw = [int(v) for v in z["meta"]]
    sd, x, fwd = build_rstd_oracle(z, torch.float64, emulate_bf16=True)
    net = ResNet([1, 1, 1, 1], feat_dim=D, image_size=hw)
    net.load_state_dict({k: (v.detach().float() if v.is_floating_point() else v) for k, v in sd.items()})
    net = net.cuda()
    c = torch.from_numpy(z["c"])
    emb = net(x.float().cuda())
    (emb * c.float().cuda()).sum().backward()
    torch.cuda.synchronize()
    want = fwd(sd, x)
    (want * c).sum().backward()
    # forward: the GPU reproduces the rounding-point model, and is as close to the reference's plain float64 output as
    # that model is (the bf16 storage of a 17-BatchNorm stack normalised over 8 samples costs ~3e-3 of cosine here)
    assert min_cos(emb.detach().cpu(), want.detach()) >= 0.9995
    assert min_cos(emb.detach().cpu(), z["emb"]) >= min_cos(want.detach(), z["emb"]) - 2e-3
    assert min_cos(emb.detach().cpu(), z["emb"]) >= 0.99
    sd32, x32, fwd32 = build_rstd_oracle(z, torch.float32, emulate_bf16=True)
    w32 = fwd32(sd32, x32)
    (w32 * c.float()).sum().backward()
    ref = {k: v.grad.numpy() for k, v in sd.items() if v.requires_grad}
    f32 = {k: v.grad.double().numpy() for k, v in sd32.items() if v.requires_grad}
    got = {k: p.grad.detach().double().cpu().numpy() for k, p in net.named_parameters()}
    assert set(got) == set(ref)                                                 # features.weight trains here
    noise_all, noise_each = _grad_errors(f32, ref)
    g_all, g_each = _grad_errors(got, ref)
    print("rstd: emb cos vs reference %.6f, gradient rel-L2 gpu %.2e (fp32-vs-fp64 band %.2e)" %
          (min_cos(emb.detach().cpu(), z["emb"]), g_all, noise_all))
    assert g_all <= max(1.5 * noise_all, 5e-3), (g_all, noise_all)
    for k, e in g_each.items():
        assert e <= 2.0 * max(noise_each[k], noise_all, 2e-3), (k, e, noise_each[k], noise_all)
    # and against the reference's own gradient norms (plain float64: the bf16 band applies)
    names = [str(n) for n in z["grad_names"]]
    gn = np.asarray([float(np.linalg.norm(got[n])) for n in names])
    big = z["grad_norms"] > 1e-2 * z["grad_norms"].max()
    np.testing.assert_allclose(gn[big], z["grad_norms"][big], rtol=max(0.1, 4 * noise_all))


def test_fp8_class_matmul_in_the_whole_step():
    """config C5's precision end to end: the same seeded step (4-block iResNet, D = 512 so that the head runs the shadow
    sweeps, 4096 slots, outlier rows included) with head_dtype "bf16" and "fp8" — the e4m3 class matmul moves the loss by
    < 5e-3 (SURVEY 8d allows 5e-2), the parameter-gradient vector by < 5 % rel-L2, and nothing in the integer state or
    the pool rows (fp32 master)."""
    from vlsfr_amd.ffc import FFC
    outs = []
    for dtype in ("bf16", "fp8"):
        torch.manual_seed(11)
        m = FFC("irtiny", 512, 4096, 32.0, "Arc", 0.5, 0.99).cuda()
        m.head_dtype = dtype
        rng = np.random.default_rng(5)
        B = 32
        ids = rng.choice(6000, size=B // 2, replace=False)
        xl = torch.from_numpy(np.concatenate([ids, rng.integers(0, 6000, B - B // 2)]).astype(np.int64))
        yl = torch.from_numpy(np.concatenate([ids, rng.integers(0, 6000, B - B // 2)]).astype(np.int64))
        losses = []
        for it in range(2):               # second step: the pool rows written by the first are swept from the shadow
            # fresh images of the same identities every step: with the SAME images (and no optimizer step in between) the
            # probe embedding equals the pool row the gallery net wrote for it, cos = 1, and ArcFace's sqrt(1 - cos^2) has no
            # clamp (ffc.py:100-103) — the forward pass is reproducible enough now (float64 statistics) to hit it
            x = common.images_from_u8(common.synth_images_u8(rng, B)).cuda()
            y = common.images_from_u8(common.synth_images_u8(rng, B)).cuda()
            m.zero_grad()
            loss = m(x, y, xl, yl)
            loss.backward()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        head = m._ensure_head()
        assert head.shadow.t.get(dtype) is not None and head.shadow.t.get("fp8" if dtype == "bf16" else "bf16") is None
        g = np.concatenate([p.grad.float().cpu().numpy().ravel() for _, p in sorted(m.probe_net.named_parameters()) if p.grad is not None])
        outs.append((losses, g, m.queue.clone(), m.lru.state_dict(), m._state().qp.copy()))
    (la, ga, qa, sa, qpa), (lb, gb, qb, sb, qpb) = outs
    np.testing.assert_allclose(lb, la, rtol=5e-3)
    assert rel_l2(gb, ga) < 5e-2, rel_l2(gb, ga)
    assert sa == sb and (qpa == qpb).all()
    assert float((qa - qb).abs().max()) < 5e-3        # pool rows = gallery embeddings of the same weights (run-to-run noise only)


def test_weight_gradients_on_a_side_stream_do_not_change_the_step():
    """vlsfr_iresnet_backward_overlap (off by default: measured slower, model/_native.py): the weight-gradient kernels of each
    backward pass on their own stream, ring of gradient buffers, event-ordered — against the default backward pass on the
    same inputs: every parameter gradient within the run-to-run noise of the fp32 atomics, integer state identical."""
    z = np.load(os.path.join(G, "step_irtiny.npz"))
    outs = []
    for overlap in (False, False, True):
        m, x, y, xl, yl = build_ffc(z, "irtiny")
        m.probe_net.overlap_wgrad = overlap
        loss = m(x, y, xl, yl)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((float(loss.detach()), m.lru.state_dict(), m._state().qp.copy(),
                     {k: p.grad.clone() for k, p in m.probe_net.named_parameters() if p.grad is not None}))
    (l0, s0, qp0, g0), (l0b, _, _, g0b), (l1, s1, qp1, g1) = outs
    cat = lambda g: np.concatenate([g[k].float().cpu().numpy().ravel() for k in sorted(g)])
    noise = rel_l2(cat(g0b), cat(g0))
    d = rel_l2(cat(g1), cat(g0))
    print("run-to-run %.2e, side-stream weight gradients vs default %.2e" % (noise, d))
    assert g0.keys() == g1.keys()
    assert d <= max(4 * noise, 5e-2)
    assert abs(l0 - l1) <= max(4 * abs(l0 - l0b), 2e-3 * abs(l0))
    assert s0 == s1 and (qp0 == qp1).all()


# (net_type, feat_dim, batch): the full-depth members of the three families that no other test builds — the reference's
# default `--net_type r50` ([3, 4, 6, 3] Bottlenecks at 224 x 224, model/resnet_std.py:242-251) and iresnet34 / iresnet200
# (model/resnet_arcface.py:167-184)
FULL_DEPTH = [("r50", 512, 4), ("ir34", 512, 8), ("ir200", 512, 4)]


@pytest.mark.parametrize("net_type,D,B", FULL_DEPTH, ids=[c[0] for c in FULL_DEPTH])
def test_full_depth_backbone_step(net_type, D, B):
    """One whole FFC step through the full-depth net: the probe embeddings of the first pass against the float64 oracle with
    bf16 storage emulation (forward only — cosine >= 0.999), the loss against the oracle's (rtol 2e-2, SURVEY
    8d), the state-dict layout against the oracle's restatement of the reference constructors, finite gradients on every
    trainable parameter, a finite SGD step, LRU / queue_position state.  (Gradient parity of these kernels and of the
    executors' wiring: tests/test_blocks_gpu.py and the emulating-oracle steps above.)"""
    from vlsfr_amd.ffc import FFC
    from vlsfr_amd.optim.fused import FusedSGD
    from oracle import ffc_ref
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    Q = 256
    o, sd, x, y, xl, yl = _emulated_oracle(net_type, D, Q, B, "Arc", 0.5, 41, None, torch.float64)
    hw = 224 if net_type == "r50" else 112
    rng = np.random.default_rng(41)
    x = common.images_from_u8(common.synth_images_u8(rng, B, hw=hw))
    y = common.images_from_u8(common.synth_images_u8(rng, B, hw=hw))
    if net_type == "r50":
        # 16 Bottlenecks whose last BatchNorm has gamma ~ 1 put a randomly initialised ReLU net at a point where bf16 storage
        # alone moves the embedding by cos 0.66 and fp32-vs-fp64 summation by cos 0.91 (scripts/diag_rstd_depth.py; the
        # executor sits ON that band, as it does at gamma x 0.1: 0.99968 vs 0.99967) — nothing can be pinned there.  Small
        # residual-branch gammas (what resnet_std.py:156-161's zero_init_residual aims at, and what trained nets have)
        # give a well-conditioned point.
        for k in sd:
            if k.endswith("bn3.weight"):
                sd[k] = sd[k] * 0.1
        o.probe = {k: (v.double().requires_grad_(bb.trainable(k, net_type)) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        o.gallery = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    m = FFC(net_type, D, Q, 32.0, "Arc", 0.5, 0.99, precise_head=True)
    assert set(m.probe_net.state_dict().keys()) == set(sd.keys())
    assert all(tuple(m.probe_net.state_dict()[k].shape) == tuple(v.shape) for k, v in sd.items())
    m.probe_net.load_state_dict(sd)
    m.gallery_net.load_state_dict(sd)
    m = m.cuda()
    m.queue.copy_(o.queue.float())
    embs = []
    hook = m.probe_net.register_forward_hook(lambda mod, i, out: embs.append(out.detach().cpu()))
    opt = FusedSGD([p for p in m.parameters() if p.requires_grad], 0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    opt.zero_grad()
    loss = m(x.cuda(), y.cuda(), xl, yl)
    loss.backward()
    hook.remove()
    torch.cuda.synchronize()
    assert embs[0].shape == (B, D) and embs[1].shape == (B, D)
    with torch.no_grad():
        emb_x = o.fwd({k: v.detach().clone() for k, v in o.probe.items()}, x.double())
        emb_32 = o.fwd({k: (v.detach().float() if v.is_floating_point() else v.clone()) for k, v in o.probe.items()}, x.float())
    c, band = min_cos(embs[0], emb_x), min_cos(emb_32, emb_x)
    print("%s full depth: embedding cosine vs the emulating float64 oracle %.6f (the oracle in fp32 vs fp64: %.6f), loss %.5f" %
          (net_type, c, band, float(loss.detach())))
    # cosine >= 0.999 — or, where 200 layers at batch 4 amplify summation-order noise beyond that, within twice
    # what an fp32 evaluation of the same rounding-point model deviates from its fp64 self
    assert 1.0 - c <= max(1e-3, 2.0 * (1.0 - band)), (c, band)
    np.testing.assert_allclose(embs[0].norm(dim=1).numpy(), 1.0, rtol=1e-5)
    assert np.isfinite(float(loss.detach()))
    n_grad = 0
    for k, p in m.probe_net.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
            n_grad += int(float(p.grad.abs().max()) > 0)
    assert n_grad >= 0.9 * sum(1 for p in m.probe_net.parameters() if p.requires_grad)
    opt.step()
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(p).all()) for p in m.probe_net.parameters())
    # bookkeeping is label-driven: identical to the oracle's, which runs both passes on the host in float64 (forward only)
    with torch.no_grad():
        want = o.forward(x.double(), y.double(), xl, yl)
    np.testing.assert_allclose(float(loss.detach()), float(want), rtol=2e-2)
    assert m.lru.state_dict() == o.lru.state_dict()
    assert m.queue_position_dict.values() == o.qp
