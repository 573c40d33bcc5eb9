"""Pins the oracle (oracle/*.py, CPU restatement) against golden vectors produced by the reference
itself (tests/golden/make_golden.py, run in the build container)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import backbones_ref as bb
from oracle import ffc_ref
from oracle.lru_ref import LRURef
from tests.golden import common

G = common.GOLDEN_DIR


def replay_lru_trace(make_lru, op_types_of):
    with open(os.path.join(G, "lru_traces.json")) as f:
        data = json.load(f)
    for case in data["cases"]:
        l = make_lru(case["capacity"])
        for op in case["ops"]:
            kind, arg, want = op[0], op[1], op[2]
            if kind == "get":
                assert l.get(arg) == want
            elif kind == "try_get":
                assert l.try_get(arg) == want
            elif kind == "view":
                assert l.view(arg) == want
            elif kind == "contains":
                assert int(arg in l) == want
            elif kind == "rollback":
                l.rollback_steps(arg)
                assert len(op_types_of(l)) == want
            else:
                assert [list(kv) for kv in l.state_dict()] == want
                assert l.cur_idx == op[3]
                assert op_types_of(l) == op[4]
    rc = data["restore_case"]
    l = make_lru(4)
    l.restore([(7, 2), (9, 0), (11, 1)])
    assert [list(kv) for kv in l.state_dict()] == rc["after_restore"]
    assert l.cur_idx == rc["cur_idx_after_restore"]
    assert l.get(5) == rc["get5"] and l.get(6) == rc["get6"]
    assert [list(kv) for kv in l.state_dict()] == rc["state_after_get"]
    l.clear()
    assert l.state_dict() == [] and l.cur_idx == rc["cur_idx_after_clear"]
    with pytest.raises(AssertionError):
        l.restore([(1, 0)])          # cur_idx != 0 after clear (lru.py:115, :132-141)


def test_lru_oracle_matches_reference_traces():
    replay_lru_trace(LRURef, lambda l: l.op_types())


HEAD_FILES = sorted(f for f in os.listdir(G) if f.startswith("head_"))


def load_head_case(fname):
    z = np.load(os.path.join(G, fname))
    Q, D, B, T, n_id, hard_neg, seed = [int(v) for v in z["meta"]]
    case = {k: z[k] for k in z.files}
    if "queue0" not in case:
        case["queue0"] = common.head_case(seed, Q, D, B, T, n_id)["queue0"]
    loss_type = fname.split("_")[1]
    return case, (Q, D, B, T, hard_neg), loss_type, float(z["hyper"][0]), float(z["hyper"][1])


@pytest.mark.parametrize("fname", HEAD_FILES)
def test_head_oracle_matches_reference(fname):
    case, (Q, D, B, T, hard_neg), loss_type, scale, margin = load_head_case(fname)
    assert hard_neg == ffc_ref.hard_neg_count(Q)
    queue = torch.from_numpy(case["queue0"]).clone()
    lru, qp = LRURef(Q), [0] * Q
    for t in range(T):
        xl, yl = case["XL"][t].tolist(), case["YL"][t].tolist()
        for s, (pl, gl, trans) in enumerate(((xl, yl, True), (yl, xl, False))):
            p = torch.from_numpy(case["P"][t, s]).clone().requires_grad_(True)
            g = torch.from_numpy(case["G"][t, s])
            loss, _ = ffc_ref.head_pass_ref(queue, lru, qp, p, g, pl, gl, trans, loss_type, scale, margin, hard_neg)
            loss.backward()
            np.testing.assert_allclose(float(loss.detach()), case["loss"][t, s], rtol=2e-5, atol=1e-5)
            np.testing.assert_allclose(p.grad.numpy(), case["dP"][t, s], rtol=2e-4, atol=2e-5)
    assert [k for k, _ in lru.state_dict()] == case["lru_keys"].tolist()
    assert [v for _, v in lru.state_dict()] == case["lru_slots"].tolist()
    assert qp == case["qp"].tolist()
    if "queue_final" in case:
        np.testing.assert_array_equal(queue.numpy(), case["queue_final"])
    else:
        np.testing.assert_array_equal(queue[:, case["touched"]].numpy(), case["queue_touched"])


def build_oracle_from_step(z, tag, dtype=torch.float64, emulate_bf16=False):
    """The oracle set to the golden case's warm state.  float64 by default: the golden step vectors
    are the reference's float64 arithmetic.  emulate_bf16: round where the device stores bf16 (oracle/backbones_ref.py)."""
    Q, D, B, seed = [int(v) for v in z["meta"]]
    layers = (1, 1, 1, 1) if tag == "irtiny" else None
    net = "irtiny" if tag == "irtiny" else "mobile"
    o = ffc_ref.FFCRef(net, D, Q, 32.0, "Arc", 0.5, 0.99, layers=layers, emulate_bf16=emulate_bf16)
    sd = common.fill_state({k: v.detach() for k, v in o.probe.items()}, seed)
    cast = lambda v: v.to(dtype) if v.is_floating_point() else v.clone()
    o.probe = {k: cast(v) for k, v in sd.items()}
    for k, v in o.probe.items():
        if bb.trainable(k):
            v.requires_grad_(True)
    o.gallery = {k: cast(v) for k, v in sd.items()}
    o.queue = torch.from_numpy(z["queue_warm"]).to(dtype)
    o.lru.restore(list(zip(z["lru_warm_keys"].tolist(), z["lru_warm_slots"].tolist())))
    o.qp = z["qp_warm"].astype(int).tolist()
    inp = common.step_inputs(seed, Q, D, B)
    x, y = common.images_from_u8(inp["xu8"]).to(dtype), common.images_from_u8(inp["yu8"]).to(dtype)
    return o, x, y, torch.from_numpy(inp["xl"]), torch.from_numpy(inp["yl"])


def sample(a):
    flat = a.reshape(-1)
    return flat[::max(1, flat.size // 4096)]


@pytest.mark.parametrize("tag", ["mobile", "irtiny"])
def test_step_oracle_matches_reference(tag):
    z = np.load(os.path.join(G, "step_%s.npz" % tag))
    o, x, y, xl, yl = build_oracle_from_step(z, tag)
    loss = o.forward(x, y, xl, yl)
    loss.backward()
    # the reference casts the logits to float32 inside the Arc/SV branch (ffc.py:97,118) even when the
    # module is float64, so agreement is bounded by one float32 rounding of the cosines
    np.testing.assert_allclose(float(loss.detach()), float(z["loss"]), rtol=1e-7)
    names = [str(n) for n in z["grad_names"]]
    gn = np.asarray([float(o.probe[n].grad.norm()) for n in names])
    np.testing.assert_allclose(gn, z["grad_norms"], rtol=1e-4, atol=1e-6)
    params = [o.probe[n] for n in names]
    ffc_ref.sgd_nesterov_step_ref(params, [p.grad for p in params], [None] * len(params), 0.1)
    for key in z.files:
        if key.startswith("grad/"):
            np.testing.assert_allclose(sample(o.probe[key[5:]].grad.numpy()), z[key], rtol=1e-3, atol=1e-5 * np.abs(z[key]).max() + 1e-9)
        elif key.startswith("after/"):
            np.testing.assert_allclose(sample(o.probe[key[6:]].detach().numpy()), z[key], rtol=1e-5, atol=2e-6)
        elif key.startswith("gallery_after/"):
            np.testing.assert_allclose(sample(o.gallery[key[14:]].numpy()), z[key], rtol=1e-12, atol=1e-13)
        elif key.startswith("buf/"):
            np.testing.assert_allclose(o.probe[key[4:]].numpy(), z[key], rtol=1e-10, atol=1e-12)
        elif key.startswith("gallery_buf/"):
            np.testing.assert_allclose(o.gallery[key[12:]].numpy(), z[key], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(o.queue.numpy(), z["queue_final"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(np.stack([z["emb_probe_x"], z["emb_probe_y"]]).shape[1:], (z["meta"][2], z["meta"][1]))
    assert [k for k, _ in o.lru.state_dict()] == z["lru_final_keys"].tolist()
    assert [v for _, v in o.lru.state_dict()] == z["lru_final_slots"].tolist()
    assert o.qp == z["qp_final"].astype(int).tolist()


def test_schedulers_match_reference_values():
    """The warm-up learning-rate schedules (optim/optimizer.py:47-128 through the factory :142-168) against lr values
    generated by the reference itself (tests/golden/make_golden.py scheduler_vectors): multistep, cosine, exponential
    and linear, with warm-up, on the grid of update(epoch, 0.0) / update(None, iter) calls main.py makes."""
    import json
    from vlsfr_amd.optim import get_optim_scheduler
    with open(os.path.join(G, "scheduler_lrs.json")) as f:
        cases = json.load(f)
    assert {c["config"]["scheduler"] for c in cases} == {"multistep", "cos", "exponential", "linear"}
    for case in cases:
        opt, sch = get_optim_scheduler([torch.nn.Parameter(torch.zeros(2))], case["config"])
        for epoch, it, lr in case["rows"]:
            if it == 0.0:
                sch.update(int(epoch), 0.0)
            else:
                sch.update(None, float(it))
            assert abs(opt.param_groups[0]["lr"] - lr) <= 1e-12 * max(1.0, abs(lr)), (case["config"]["scheduler"], epoch, it)


def build_rstd_oracle(z, dtype=torch.float64, emulate_bf16=False):
    """The oracle's torchvision-style ResNet on the seeded state / images of tests/golden/backbone_rstd.npz."""
    from oracle import backbones_ref as bb
    D, B, seed, hw = [int(v) for v in z["meta"]]
    sd0, _ = bb.make_backbone("rtiny", D)                    # names / shapes only; 224 x 224 -> fc is 2048 * 49 wide
    sd0 = bb.resnet_std_state((1, 1, 1, 1), D, None, hw)
    sd = common.fill_state(sd0, seed)
    sd = {k: (v.to(dtype).requires_grad_(bb.trainable(k, "rtiny")) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    rng = np.random.default_rng(seed)
    x = common.images_from_u8(common.synth_images_u8(rng, B, hw=hw)).to(dtype)
    return sd, x, (lambda s, xx: bb.resnet_std_forward(s, xx, (1, 1, 1, 1), emulate_bf16))


def test_resnet_std_oracle_matches_reference():
    """The oracle's restatement of model/resnet_std.py (Bottleneck ResNet: 7x7/2 stem, max-pool, post-add ReLU, trainable
    features.weight) against float64 outputs of the reference's own class: embeddings, every gradient norm, sampled
    gradient tensors, BatchNorm running means."""
    z = np.load(os.path.join(G, "backbone_rstd.npz"))
    sd, x, fwd = build_rstd_oracle(z)
    emb = fwd(sd, x)
    (emb * torch.from_numpy(z["c"])).sum().backward()
    np.testing.assert_allclose(emb.detach().numpy(), z["emb"], rtol=1e-9, atol=1e-12)
    names = [str(n) for n in z["grad_names"]]
    np.testing.assert_allclose([float(sd[n].grad.norm()) for n in names], z["grad_norms"], rtol=1e-7)
    for key in z.files:
        if key.startswith("grad/"):
            np.testing.assert_allclose(sample(sd[key[5:]].grad.numpy()), z[key], rtol=1e-5, atol=1e-9)
        elif key.startswith("buf/"):
            np.testing.assert_allclose(sd[key[4:]].numpy(), z[key], rtol=1e-6, atol=1e-9)
