"""GPU parity of the fused DCP head (through the C-ABI) against the reference's golden vectors, the
oracle on seeded inputs, and size-independent properties at the 1M-identity configuration."""
import os

import numpy as np
import pytest
import torch

from oracle import ffc_ref
from oracle.lru_ref import LRURef
from tests.golden import common
from tests.test_oracle_golden import HEAD_FILES, load_head_case

pytestmark = pytest.mark.gpu

# Tolerances (stated per SURVEY §8d): "precise" = split-bf16 MFMA products (~2^-16 relative per
# product) -> fp32-class agreement; plain bf16 operands -> 2^-9 relative per product.
TOL = {True: dict(loss_rtol=2e-5, loss_atol=2e-5, dp_rtol=1e-3, dp_atol=2e-5),
       False: dict(loss_rtol=3e-3, loss_atol=3e-3, dp_rtol=5e-2, dp_atol=1e-2)}


def make_head(queue0, loss_type, scale, margin, precise, n_chunks=0):
    from vlsfr_amd.head import DcpHead
    q = torch.from_numpy(queue0).cuda().contiguous()
    return DcpHead(q, scale, margin, loss_type, precise=precise, n_chunks=n_chunks)


@pytest.mark.parametrize("precise", [True, False])
@pytest.mark.parametrize("fname", HEAD_FILES)
def test_head_matches_reference_golden(fname, precise):
    case, (Q, D, B, T, hard_neg), loss_type, scale, margin = load_head_case(fname)
    head = make_head(case["queue0"], loss_type, scale, margin, precise)
    assert head.hard_neg == hard_neg
    tol = TOL[precise]
    for t in range(T):
        xl, yl = case["XL"][t], case["YL"][t]
        for s, (pl, gl, trans) in enumerate(((xl, yl, True), (yl, xl, False))):
            p = torch.from_numpy(case["P"][t, s]).cuda().requires_grad_(True)
            g = torch.from_numpy(case["G"][t, s]).cuda()
            loss = head.run_pass(p, g, pl, gl, trans)
            loss.backward()
            np.testing.assert_allclose(float(loss), case["loss"][t, s], rtol=tol["loss_rtol"], atol=tol["loss_atol"],
                                       err_msg="%s t=%d pass=%d" % (fname, t, s))
            np.testing.assert_allclose(p.grad.cpu().numpy(), case["dP"][t, s], rtol=tol["dp_rtol"],
                                       atol=tol["dp_atol"] * max(1.0, np.abs(case["dP"][t, s]).max()),
                                       err_msg="%s t=%d pass=%d" % (fname, t, s))
    torch.cuda.synchronize()
    st = head.lru.state_dict()
    assert [k for k, _ in st] == case["lru_keys"].tolist()
    assert [v for _, v in st] == case["lru_slots"].tolist()
    assert head.qp.tolist() == case["qp"].tolist()
    qf = head.queue.cpu().numpy()
    if "queue_final" in case:
        np.testing.assert_array_equal(qf, case["queue_final"])      # pool rows are bitwise copies of g
    else:
        np.testing.assert_array_equal(qf[:, case["touched"]], case["queue_touched"])


def oracle_pass(queue, lru, qp, p, g, pl, gl, trans, loss_type, scale, margin, hard_neg):
    pt = torch.from_numpy(p).double().requires_grad_(True)
    loss, _ = ffc_ref.head_pass_ref(queue, lru, qp, pt, torch.from_numpy(g).double(), pl.tolist(), gl.tolist(), trans,
                                    loss_type, scale, margin, hard_neg)
    loss.backward()
    return float(loss.detach()), pt.grad.numpy()


@pytest.mark.parametrize("loss_type,margin", [("Arc", 0.5), ("AM", 0.4), ("SV", 0.35)])
@pytest.mark.parametrize("Q,D,B,n_id", [(5000, 128, 64, 4000), (3001, 512, 40, 9000), (70000, 512, 96, 50000)])
def test_head_vs_oracle_seeded(loss_type, margin, Q, D, B, n_id):
    """Larger pools, ragged Q (not a multiple of the 32-column tile), B not a multiple of 16 / above
    one 64-row block, outlier rows (n_id > Q) — against the float64 oracle."""
    T = 3
    case = common.head_case(1000 + Q + B, Q, D, B, T, n_id)
    for precise in (True, False):
        tol = TOL[precise]
        head = make_head(case["queue0"], loss_type, 32.0, margin, precise)
        queue = torch.from_numpy(case["queue0"]).double()
        lru, qp = LRURef(Q), [0] * Q
        for t in range(T):
            xl, yl = case["XL"][t], case["YL"][t]
            for s, (pl, gl, trans) in enumerate(((xl, yl, True), (yl, xl, False))):
                want_loss, want_dp = oracle_pass(queue, lru, qp, case["P"][t, s], case["G"][t, s], pl, gl, trans,
                                                 loss_type, 32.0, margin, head.hard_neg)
                p = torch.from_numpy(case["P"][t, s]).cuda().requires_grad_(True)
                loss = head.run_pass(p, torch.from_numpy(case["G"][t, s]).cuda(), pl, gl, trans)
                loss.backward()
                np.testing.assert_allclose(float(loss), want_loss, rtol=tol["loss_rtol"], atol=tol["loss_atol"])
                np.testing.assert_allclose(p.grad.cpu().numpy(), want_dp, rtol=tol["dp_rtol"],
                                           atol=tol["dp_atol"] * max(1.0, np.abs(want_dp).max()))
        assert head.lru.state_dict() == lru.state_dict()
        assert head.qp.tolist() == qp
        np.testing.assert_array_equal(head.queue.cpu().numpy(), queue.float().numpy())


def test_head_full_size_properties():
    """BASELINE config C2 head: Q = 1M identities, D = 512, B = 64.  The oracle does not finish in
    seconds here, so check size-independent properties: (1) the result does not depend on the
    column partition (n_chunks), (2) the rollback pass leaves pool, LRU and queue_position state
    bit-identical, (3) everything is finite.  test_head_dense_torch_reference_1m adds a dense fp32
    evaluation of the same configuration."""
    from vlsfr_amd.head import DcpHead
    Q, D, B = 1 << 20, 512, 64
    gen = torch.Generator(device="cuda").manual_seed(5)
    queue0 = torch.nn.functional.normalize(torch.randn(2, Q, D, device="cuda", generator=gen), dim=2)

    def run(n_chunks):
        rng = np.random.default_rng(3)
        head = DcpHead(queue0.clone(), 32.0, 0.5, "Arc", precise=False, n_chunks=n_chunks)
        ids = rng.choice(Q * 2, size=B, replace=False).astype(np.int64)
        warm = rng.choice(Q * 2, size=4096, replace=False).astype(np.int64)
        warm[:B // 2] = ids[:B // 2]
        with torch.no_grad():
            for s in range(0, 4096, 512):   # commit some identities so hits / ones_idx / positives occur
                w = warm[s:s + 512]
                head.run_pass(torch.from_numpy(common.unit_rows(rng, 512, D)).cuda(),
                              torch.from_numpy(common.unit_rows(rng, 512, D)).cuda(), w, w, False)
        before_q = head.queue.clone()
        before_lru, before_qp = head.lru._state_arrays(), head.qp.copy()
        p = torch.from_numpy(common.unit_rows(rng, B, D)).cuda().requires_grad_(True)
        loss = head.run_pass(p, torch.from_numpy(common.unit_rows(rng, B, D)).cuda(), ids, ids, True)
        loss.backward()
        assert torch.equal(head.queue, before_q)
        after_lru = head.lru._state_arrays()
        assert np.array_equal(before_lru[0], after_lru[0]) and np.array_equal(before_lru[1], after_lru[1])
        assert np.array_equal(before_qp, head.qp)
        return float(loss), p.grad.cpu().numpy()

    l0, d0 = run(0)
    l1, d1 = run(128)
    assert np.isfinite(l0) and np.isfinite(d0).all()
    np.testing.assert_allclose(l0, l1, rtol=1e-4)
    np.testing.assert_allclose(d0, d1, rtol=2e-2, atol=2e-3 * np.abs(d0).max())


def test_head_dense_torch_reference_1m():
    """Same configuration against a dense fp32 PyTorch evaluation on the GPU (F.linear + CE)."""
    Q, D, B = 1 << 20, 512, 64
    gen = torch.Generator(device="cuda").manual_seed(11)
    queue0 = torch.nn.functional.normalize(torch.randn(2, Q, D, device="cuda", generator=gen), dim=2)
    from vlsfr_amd.head import DcpHead
    head = DcpHead(queue0.clone(), 32.0, 0.5, "Arc", precise=False)
    rng = np.random.default_rng(8)
    labels = rng.choice(Q, size=B, replace=False).astype(np.int64)
    # pre-fill the LRU so that label k owns slot k (restore takes MRU->LRU (key, slot) pairs)
    head.lru.restore([(int(k), int(k)) for k in range(4096)])
    labels = rng.choice(4096, size=B, replace=False).astype(np.int64)
    p = torch.from_numpy(common.unit_rows(rng, B, D)).cuda().requires_grad_(True)
    g = torch.from_numpy(common.unit_rows(rng, B, D)).cuda()
    loss = head.run_pass(p, g, labels, labels, True)
    loss.backward()
    # dense reference: all labels hit -> rows = qp = 0 -> g written to row 0; ones_idx = all label slots,
    # so variant 2 reads queue[1] at those slots.
    W = queue0.clone()
    lab = torch.from_numpy(labels).cuda()
    W[0, lab] = g
    pr = p.detach().clone().requires_grad_(True)
    w2 = W[0].clone()
    w2[lab] = W[1, lab]
    tot = 0
    for w in (W[0], w2):
        cos = pr @ w.t()
        gt = cos.gather(1, lab.view(-1, 1))
        new = gt * np.cos(0.5) - torch.sqrt(1 - gt * gt) * np.sin(0.5)
        cos = cos.scatter(1, lab.view(-1, 1), new)
        tot = tot + torch.nn.functional.cross_entropy(cos * 32.0, lab)
    tot.backward()
    np.testing.assert_allclose(float(loss), float(tot), rtol=2e-3)
    np.testing.assert_allclose(p.grad.cpu().numpy(), pr.grad.cpu().numpy(), rtol=5e-2,
                               atol=4e-3 * float(pr.grad.abs().max()))


@pytest.mark.parametrize("loss_type,margin", [("Arc", 0.5), ("SV", 0.35)])
def test_head_row_sharded_equals_full_batch(loss_type, margin):
    """Data-parallel semantics (parallel.py): two simulated ranks, each with its own replica of the
    pool and allocator, each holding half of the probe rows but seeing the whole batch's gallery
    embeddings and labels.  Sum of the rank losses == single-process loss on the concatenated batch,
    dL/dp rows match, and both replicas end in the single-process pool / LRU / queue_position state."""
    Q, D, B, T, n_id = 4000, 128, 48, 3, 5000
    case = common.head_case(4242, Q, D, B, T, n_id)
    full = make_head(case["queue0"], loss_type, 32.0, margin, True)
    ranks = [make_head(case["queue0"], loss_type, 32.0, margin, True) for _ in range(2)]
    h = B // 2
    for t in range(T):
        xl, yl = case["XL"][t], case["YL"][t]
        for s, (pl, gl, trans) in enumerate(((xl, yl, True), (yl, xl, False))):
            g = torch.from_numpy(case["G"][t, s]).cuda()
            p = torch.from_numpy(case["P"][t, s]).cuda().requires_grad_(True)
            loss = full.run_pass(p, g, pl, gl, trans)
            loss.backward()
            tot, grads = 0.0, []
            for r, head in enumerate(ranks):
                pr = torch.from_numpy(case["P"][t, s][r * h:(r + 1) * h]).cuda().requires_grad_(True)
                lr = head.run_pass(pr, g, pl, gl, trans, row_offset=r * h)
                lr.backward()
                tot += float(lr.detach())
                grads.append(pr.grad)
            np.testing.assert_allclose(tot, float(loss.detach()), rtol=1e-5)
            np.testing.assert_allclose(torch.cat(grads).cpu().numpy(), p.grad.cpu().numpy(), rtol=1e-4, atol=1e-6)
    for head in ranks:
        assert head.lru.state_dict() == full.lru.state_dict()
        assert head.qp.tolist() == full.qp.tolist()
        assert torch.equal(head.queue, full.queue)


class _SimComm(object):
    """In-process stand-in for the three collectives of the sharded head: the simulated ranks run
    phase by phase, and each collective is evaluated over the list of all ranks' tensors."""

    def __init__(self, world):
        self.world, self.bufs = world, {}

    def run(self, name, tensors):
        if name == "max":
            out = torch.stack(tensors).max(0).values
        elif name == "sum":
            out = torch.stack(tensors).sum(0)
        else:
            out = torch.stack(tensors)
        return [out.clone() for _ in tensors]


def _sharded_pass(shards, p, g, pl, gl, trans):
    """Drives `world` simulated ranks through partial -> combine -> finish in lockstep."""
    world = len(shards)
    sts = [h.begin(p, g, pl, gl, trans) for h in shards]
    if sts[0]["thr"] is not None:                     # SV: all-reduce(max) of the hard-example thresholds
        thr = torch.stack([s["thr"] for s in sts]).max(0).values
        for s in sts:
            s["thr"] = thr.clone()
    sts = [h.sweep(s) for h, s in zip(shards, sts)]
    sim = _SimComm(world)

    class Comm(object):            # replays one rank's combine() against pre-computed collective results
        def __init__(self, r):
            self.r, self.step = r, 0

        def all_reduce_max(self, t):
            return torch.stack([s["M"] for s in sts]).max(0).values

        def all_gather(self, t):
            key = "cand_val" if t.dtype == torch.float32 else "cand_col"
            return torch.stack([s[key] for s in sts])

        def all_reduce_sum(self, t):
            self.mine = t
            return None
    comms = [Comm(r) for r in range(world)]
    for r, h in enumerate(shards):           # first: everything up to the sum-reduce (needs all ranks' packed)
        h.combine(sts[r], comms[r])
    total = torch.stack([c.mine for c in comms]).sum(0)
    outs = []
    for r, h in enumerate(shards):
        sts[r]["packed"] = total.clone()
        outs.append(h.finish(sts[r]))
    return outs


@pytest.mark.parametrize("loss_type,margin,n_id", [("Arc", 0.5, 1500), ("AM", 0.4, 6000), ("SV", 0.35, 1500), ("SV", 0.35, 6000)])
@pytest.mark.parametrize("world", [2, 4])
def test_identity_sharded_head_equals_single_pool(loss_type, margin, n_id, world):
    """The identity-sharded pool (each rank owns Q / world slots, softmax state combined with
    all-reduce(max) / all-reduce(sum)) reproduces the single-pool head: loss, dL/dp for every row,
    pool contents, LRU and queue_position state — including evictions and outlier rows (n_id > Q)."""
    from vlsfr_amd.head import DcpHead, ShardedDcpHead
    Q, D, B, T = 2048, 128, 40, 3
    case = common.head_case(777 + world, Q, D, B, T, n_id)
    full = make_head(case["queue0"], loss_type, 32.0, margin, True)
    q0 = torch.from_numpy(case["queue0"]).cuda()
    Qs = Q // world
    shards = [ShardedDcpHead(q0[:, r * Qs:(r + 1) * Qs].contiguous(), r, world, Q, 32.0, margin, loss_type, precise=True)
              for r in range(world)]
    for t in range(T):
        xl, yl = case["XL"][t], case["YL"][t]
        for s, (pl, gl, trans) in enumerate(((xl, yl, True), (yl, xl, False))):
            g = torch.from_numpy(case["G"][t, s]).cuda()
            p = torch.from_numpy(case["P"][t, s]).cuda().requires_grad_(True)
            loss = full.run_pass(p, g, pl, gl, trans)
            loss.backward()
            outs = _sharded_pass(shards, p.detach(), g, pl, gl, trans)
            for l, dP in outs:
                np.testing.assert_allclose(float(l), float(loss.detach()), rtol=2e-5, atol=1e-5)
                np.testing.assert_allclose(dP.cpu().numpy(), p.grad.cpu().numpy(), rtol=2e-4, atol=2e-5)
    whole = torch.cat([h.queue for h in shards], dim=1)
    assert torch.equal(whole, full.queue)
    for h in shards:
        assert h.lru.state_dict() == full.lru.state_dict() and h.qp.tolist() == full.qp.tolist()
