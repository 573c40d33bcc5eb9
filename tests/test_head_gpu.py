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
@pytest.mark.parametrize("Q,D,B,n_id", [(5000, 128, 64, 4000), (3001, 512, 40, 9000), (70000, 512, 96, 50000),
                                        (70000, 512, 256, 50000), (66000, 512, 256, 90000)])
def test_head_vs_oracle_seeded(loss_type, margin, Q, D, B, n_id):
    """Larger pools, ragged Q (not a multiple of the 32-column tile), B not a multiple of 16 / above
    one 64-row block, the benchmarked batch_size 256 (four row blocks), outlier rows (n_id > Q) — against the
    float64 oracle.  With D = 512 the non-precise head runs the bf16-shadow sweep (csrc/head16.hip)."""
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
        sts[r]["summed"] = total.clone()
        l, dP = h.finish(sts[r])
        outs.append((l.clone(), dP))
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


def test_shadow_sweep_matches_fp32_stream_and_tracks_the_pool(monkeypatch):
    """The bf16 shadow of queue[0] (D = 512, plain bf16 operands): (1) the LDS-DMA sweep over it gives the fp32-streaming
    kernel's result (same bf16 roundings of W and P; only the summation order and the fixed softmax reference exponent
    differ), for AM / Arc / SV with outlier rows; (2) after committing passes the shadow IS bf16(queue[0]) bit for bit;
    (3) a torch-side write to the pool (copy_, as load_state_dict does) rebuilds it."""
    Q, D, B, T, n_id = 40000, 512, 200, 2, 60000
    case = common.head_case(99, Q, D, B, T, n_id)
    for loss_type, margin in (("Arc", 0.5), ("AM", 0.4), ("SV", 0.35)):
        monkeypatch.setenv("VLSFR_HEAD_SHADOW", "1")
        fast = make_head(case["queue0"], loss_type, 32.0, margin, False)
        monkeypatch.setenv("VLSFR_HEAD_SHADOW", "0")
        slow = make_head(case["queue0"], loss_type, 32.0, margin, False)
        for t in range(T):
            xl, yl = case["XL"][t], case["YL"][t]
            for s_, (pl, gl, trans) in enumerate(((xl, yl, True), (yl, xl, False))):
                out = []
                for head, flag in ((fast, "1"), (slow, "0")):
                    monkeypatch.setenv("VLSFR_HEAD_SHADOW", flag)
                    p = torch.from_numpy(case["P"][t, s_]).cuda().requires_grad_(True)
                    loss = head.run_pass(p, torch.from_numpy(case["G"][t, s_]).cuda(), pl, gl, trans)
                    loss.backward()
                    out.append((float(loss), p.grad.cpu().numpy()))
                np.testing.assert_allclose(out[0][0], out[1][0], rtol=2e-5)
                np.testing.assert_allclose(out[0][1], out[1][1], rtol=2e-3, atol=2e-4 * np.abs(out[1][1]).max())
        assert fast.shadow.t.get('bf16') is not None and slow.shadow.t.get('bf16') is None
        assert torch.equal(fast.shadow.t['bf16'], fast.queue[0].to(torch.bfloat16))
        assert torch.equal(fast.queue, slow.queue)
    monkeypatch.setenv("VLSFR_HEAD_SHADOW", "1")
    rng = np.random.default_rng(5)
    fast.queue.copy_(torch.from_numpy(common.unit_rows(rng, 2, Q, D)).cuda())
    p = torch.from_numpy(case["P"][0, 0]).cuda()
    fast.run_pass(p, torch.from_numpy(case["G"][0, 0]).cuda(), case["XL"][0], case["YL"][0], True)
    assert torch.equal(fast.shadow.t['bf16'], fast.queue[0].to(torch.bfloat16))
    # (4) a write torch's version counter cannot see (through .data, as an in-place broadcast does): stale until invalidate()
    fast.queue.data.copy_(torch.from_numpy(common.unit_rows(rng, 2, Q, D)).cuda())
    fast.run_pass(p, torch.from_numpy(case["G"][0, 0]).cuda(), case["XL"][0], case["YL"][0], True)
    assert not torch.equal(fast.shadow.t['bf16'], fast.queue[0].to(torch.bfloat16))
    fast.shadow.invalidate()
    loss_a = float(fast.run_pass(p, torch.from_numpy(case["G"][0, 0]).cuda(), case["XL"][0], case["YL"][0], True))
    assert torch.equal(fast.shadow.t['bf16'], fast.queue[0].to(torch.bfloat16))
    monkeypatch.setenv("VLSFR_HEAD_SHADOW", "0")
    ref = make_head(fast.queue.cpu().numpy(), "SV", 32.0, 0.35, False)
    ref.lru, ref.qp = fast.lru, fast.qp
    loss_b = float(ref.run_pass(p, torch.from_numpy(case["G"][0, 0]).cuda(), case["XL"][0], case["YL"][0], True))
    np.testing.assert_allclose(loss_a, loss_b, rtol=2e-5)


def _shadow_matches_pool(head, dtype, queue0):
    """The sweep's mirror of queue[0] against a fresh conversion of the fp32 master, in slabs (10 M slots: 10.7 GB)."""
    sh = head.shadow.t[dtype]
    Q = queue0.shape[1]
    if dtype == "bf16":
        for a0 in range(0, Q, 1 << 20):
            if not torch.equal(sh[a0:a0 + (1 << 20)], queue0[0, a0:a0 + (1 << 20)].to(torch.bfloat16)):
                return False
        return True
    per = 8192                                                      # tiles of 128 slots per slab
    nt = (Q + 127) // 128
    b = sh.view(nt, 131072)
    for t0 in range(0, nt, per):
        t1 = min(t0 + per, nt)
        Rm, Tm = _decode_shadow8(b[t0:t1].reshape(-1), (t1 - t0) * 128)
        rows = queue0[0, t0 * 128:min(t1 * 128, Q)]
        want = torch.zeros((t1 - t0) * 128, 512, dtype=torch.uint8, device=sh.device)
        want[:rows.shape[0]] = (rows * 64.0).to(torch.float8_e4m3fn).view(torch.uint8)
        if not (torch.equal(Rm, want) and torch.equal(Tm, want)):
            return False
    return True


@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_head_metric_size_properties(dtype):
    """The metric's own head: Q = 10 485 760 identities, D = 512, batch_size 256 (BASELINE.json `metric`), with the bf16
    sweep and with config C5's e4m3 sweep (10.7 GB fragment-major shadow: byte offsets beyond 2^32).  The oracle
    cannot run here; size-independent properties instead: the result does not depend on the column partition
    (n_chunks 0 / 256), the rollback pass leaves pool, shadow, LRU and queue_position bit-identical, the commit pass
    changes exactly the written rows (fp32 master and shadow), everything is finite, and a dense fp32 PyTorch
    evaluation of 64 of the rows over the whole pool agrees (loss rtol 2e-3 bf16 / 5e-3 fp8, SURVEY 8d: 2e-2 / 5e-2)."""
    from vlsfr_amd.ffc import build_pool
    from vlsfr_amd.head import DcpHead
    Q, D, B = 10 << 20, 512, 256
    queue0 = build_pool(Q, D, "cuda", seed=7)
    rng = np.random.default_rng(3)
    ar = np.arange(Q)
    labels = rng.choice(Q, size=B, replace=False).astype(np.int64)
    p_np, g_np = common.unit_rows(rng, B, D), common.unit_rows(rng, B, D)
    res = []
    for n_chunks in (0, 256):
        head = DcpHead(queue0, 32.0, 0.5, "Arc", precise=False, n_chunks=n_chunks)
        head.head_dtype = dtype
        head.lru.restore_arrays(ar.astype(np.int64), ar.astype(np.int32))
        before_lru, before_qp = head.lru.state_arrays(), head.qp.copy()
        p = torch.from_numpy(p_np).cuda().requires_grad_(True)
        g = torch.from_numpy(g_np).cuda()
        loss = head.run_pass(p, g, labels, labels, True)
        loss.backward()
        torch.cuda.synchronize()
        assert head.shadow.t.get(dtype) is not None and len(head.shadow.t) == 1
        assert _shadow_matches_pool(head, dtype, queue0)          # rollback: pool untouched, shadow == cast(pool)
        after = head.lru.state_arrays()
        assert np.array_equal(before_lru[0], after[0]) and np.array_equal(before_lru[1], after[1])
        assert np.array_equal(before_qp, head.qp)
        res.append((float(loss), p.grad.cpu().numpy()))
        del head
    (l0, d0), (l1, d1) = res
    assert np.isfinite(l0) and np.isfinite(d0).all()
    np.testing.assert_allclose(l0, l1, rtol=1e-5 if dtype == "bf16" else 1e-4)
    np.testing.assert_allclose(d0, d1, rtol=2e-3, atol=2e-4 * np.abs(d0).max())
    # dense fp32 reference for the first 64 rows: every label hits (full residency), rows = qp = 0 -> g goes to row 0
    # of its slot and variant 2 reads queue[1] there (ones_idx = all label slots)
    lab = torch.from_numpy(labels).cuda()
    sub = slice(0, 64)
    pr = torch.from_numpy(p_np).cuda()[sub].clone().requires_grad_(True)
    g = torch.from_numpy(g_np).cuda()
    tot = 0
    for v in range(2):
        cos = torch.empty(64, Q, device="cuda")
        for a0 in range(0, Q, 1 << 21):
            cos[:, a0:a0 + (1 << 21)] = pr.detach() @ queue0[0, a0:a0 + (1 << 21)].t()
        cos[:, lab] = pr.detach() @ (g if v == 0 else queue0[1, lab]).t()
        # loss and gradient of this variant, columns other than the targets treated as constants of p through W
        z = cos * 32.0
        gt = cos.gather(1, lab[sub].view(-1, 1))
        new = gt * np.cos(0.5) - torch.sqrt(1 - gt * gt) * np.sin(0.5)
        z.scatter_(1, lab[sub].view(-1, 1), new * 32.0)
        lse = torch.logsumexp(z, dim=1)
        tot = tot + float((lse - z.gather(1, lab[sub].view(-1, 1)).view(-1)).sum()) / B
        del cos, z
    # the 64-row share of the loss is not separable from the kernel's scalar, so compare through a second run on those rows
    head = DcpHead(queue0, 32.0, 0.5, "Arc", precise=False)
    head.head_dtype = dtype
    head.lru.restore_arrays(ar.astype(np.int64), ar.astype(np.int32))
    p64 = torch.from_numpy(p_np[:64]).cuda().requires_grad_(True)
    # same special columns as the full batch: all 256 gallery rows are written, only 64 probe rows are evaluated
    l64 = head.run_pass(p64, g, labels, labels, True, row_offset=0)
    np.testing.assert_allclose(float(l64), tot, rtol=2e-3 if dtype == "bf16" else 5e-3)
    # the commit pass changes exactly the written rows of the fp32 master and of the shadow
    want = queue0.clone()                  # (head.queue IS queue0)
    want[0, lab] = g                       # every label hits with queue_position 0: row 0 of its slot takes g
    head.run_pass(torch.from_numpy(p_np).cuda(), g, labels, labels, False)
    torch.cuda.synchronize()
    assert torch.equal(head.queue, want)
    assert _shadow_matches_pool(head, dtype, want)
    assert head.qp[labels].tolist() == [1] * B and int(head.qp.sum()) == B


@pytest.mark.parametrize("loss_type,margin,n_id", [("Arc", 0.5, 12000), ("SV", 0.35, 40000)])
def test_identity_sharded_head_world8_bf16_shadow(loss_type, margin, n_id):
    """Eight simulated ranks (the node size the metric is quoted on), D = 512, plain bf16 operands: every rank sweeps
    the bf16 shadow of ITS slots for all rows with the LDS-DMA kernel; the combined result equals the single-pool head
    (which runs the same kernel over the whole pool) up to summation order."""
    from vlsfr_amd.head import ShardedDcpHead
    Q, D, B, T, world = 32768, 512, 64, 2, 8
    case = common.head_case(4100, Q, D, B, T, n_id)
    full = make_head(case["queue0"], loss_type, 32.0, margin, False)
    q0 = torch.from_numpy(case["queue0"]).cuda()
    Qs = Q // world
    shards = [ShardedDcpHead(q0[:, r * Qs:(r + 1) * Qs].contiguous(), r, world, Q, 32.0, margin, loss_type, precise=False)
              for r in range(world)]
    for t in range(T):
        xl, yl = case["XL"][t], case["YL"][t]
        for s_, (pl, gl, trans) in enumerate(((xl, yl, True), (yl, xl, False))):
            g = torch.from_numpy(case["G"][t, s_]).cuda()
            p = torch.from_numpy(case["P"][t, s_]).cuda().requires_grad_(True)
            loss = full.run_pass(p, g, pl, gl, trans)
            loss.backward()
            for l, dP in _sharded_pass(shards, p.detach(), g, pl, gl, trans):
                np.testing.assert_allclose(float(l), float(loss.detach()), rtol=5e-5, atol=1e-5)
                np.testing.assert_allclose(dP.cpu().numpy(), p.grad.cpu().numpy(), rtol=2e-3,
                                           atol=2e-4 * float(p.grad.abs().max()))
    assert all(h.shadow.t.get('bf16') is not None for h in shards)
    assert torch.equal(torch.cat([h.queue for h in shards], dim=1), full.queue)
    assert torch.equal(torch.cat([h.shadow.t['bf16'] for h in shards], dim=0), full.shadow.t['bf16'])
    for h in shards:
        assert h.lru.state_dict() == full.lru.state_dict() and h.qp.tolist() == full.qp.tolist()


# ---- fp8 (OCP e4m3) sweep: csrc/head8.hip, the fp8 slice of config C5 (SURVEY 8d tolerances: cos >= 0.99, loss rtol 5e-2) ----
def _decode_shadow8(buf, Q):
    """The fragment-major fp8 shadow (csrc/head8.hip header) back to two [ceil(Q/128)*128, 512] byte matrices: the R part
    (operand of the first product) and the T part (the transposed tile, operand of the second)."""
    nt = (Q + 127) // 128
    b = buf.view(nt, 131072)
    R = b[:, :65536].reshape(nt, 8, 4, 2, 4, 16, 16)            # [tile][mb][ks][half][g][m][byte]
    Rm = R.permute(0, 1, 5, 2, 4, 3, 6).reshape(nt * 128, 512)  # row 16 mb + m, feature 128 ks + 32 g + 16 half + byte
    T = b[:, 65536:].reshape(nt, 32, 2, 4, 16, 4, 4)            # [tile][nb][half][g][n][byte >> 2][byte & 3]
    Tm = T.permute(0, 2, 5, 3, 6, 1, 4).reshape(nt * 128, 512)  # row 16 (4 half + (byte >> 2)) + 4 g + (byte & 3), feature 16 nb + n
    return Rm, Tm


def _expected_fp8(queue0_dev, Q):
    want = torch.zeros((Q + 127) // 128 * 128, 512, dtype=torch.uint8, device=queue0_dev.device)
    want[:Q] = (queue0_dev * 64.0).to(torch.float8_e4m3fn).view(torch.uint8)
    return want


def test_fp8_shadow_layout_and_updates():
    """vlsfr_pool_shadow8_build writes e4m3(64 x) of every pool row into both fragment-major parts (ragged Q: the rows past
    the pool are zeros); after committing passes the image equals a fresh build of the updated pool, byte for byte."""
    Q, B = 3001, 40
    case = common.head_case(77, Q, 512, B, 2, 9000)
    head = make_head(case["queue0"], "Arc", 32.0, 0.5, False)
    head.head_dtype = "fp8"
    assert head.shadow.ptr(True, "fp8") is not None
    Rm, Tm = _decode_shadow8(head.shadow.t["fp8"], Q)
    want = _expected_fp8(head.queue[0], Q)
    assert torch.equal(Rm, want) and torch.equal(Tm, want)
    for t in range(2):
        xl, yl = case["XL"][t], case["YL"][t]
        for s, (pl, gl, trans) in enumerate(((xl, yl, True), (yl, xl, False))):
            p = torch.from_numpy(case["P"][t, s]).cuda().requires_grad_(True)
            head.run_pass(p, torch.from_numpy(case["G"][t, s]).cuda(), pl, gl, trans).backward()
    Rm, Tm = _decode_shadow8(head.shadow.t["fp8"], Q)
    want2 = _expected_fp8(head.queue[0], Q)
    assert not torch.equal(want2, want)                       # the commits wrote rows
    assert torch.equal(Rm, want2) and torch.equal(Tm, want2)
    assert head.shadow.t.get("bf16") is None                  # the fp8 head never builds the bf16 mirror


@pytest.mark.parametrize("loss_type,margin", [("Arc", 0.5), ("AM", 0.4), ("SV", 0.35)])
@pytest.mark.parametrize("Q,B,n_id", [(3001, 40, 9000), (70000, 256, 50000), (66000, 200, 90000)])
def test_head_fp8_vs_oracle(loss_type, margin, Q, B, n_id):
    """The e4m3 sweep against the float64 oracle: ragged Q (not a multiple of the 128-column tile), B below / above one
    128-row block and not a multiple of 16, outlier rows (n_id > Q: hard-negative top-k).  Bookkeeping and pool rows are
    exact (the fp32 master is what is written); loss within 5e-3 (SURVEY: 5e-2), every dL/dp row within cos 0.99 of the
    oracle's and the whole dL/dp within 8 % rel-L2 — the fp8 rounding of pool, probe and numerators."""
    T, D = 3, 512
    case = common.head_case(3000 + Q + B, Q, D, B, T, n_id)
    head = make_head(case["queue0"], loss_type, 32.0, margin, False)
    head.head_dtype = "fp8"
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
            np.testing.assert_allclose(float(loss), want_loss, rtol=5e-3)
            got = p.grad.double().cpu().numpy()
            num = (got * want_dp).sum(1)
            den = np.linalg.norm(got, axis=1) * np.linalg.norm(want_dp, axis=1)
            live = den > 1e-12 * max(den.max(), 1e-30)
            assert (num[live] / den[live]).min() > 0.99
            assert np.linalg.norm(got - want_dp) <= 0.08 * np.linalg.norm(want_dp)
    assert head.shadow.t.get("fp8") is not None
    assert head.lru.state_dict() == lru.state_dict()
    assert head.qp.tolist() == qp
    np.testing.assert_array_equal(head.queue.cpu().numpy(), queue.float().numpy())


@pytest.mark.parametrize("loss_type,margin,n_id", [("Arc", 0.5, 12000), ("SV", 0.35, 40000)])
def test_identity_sharded_head_world8_fp8(loss_type, margin, n_id):
    """C5's precision on the node size the metric is quoted on: eight simulated ranks, each sweeping the e4m3 shadow of ITS
    slots for all rows; the combined result equals the single-pool fp8 head up to summation order (shards are tile-aligned,
    so every numerator is quantised against the same half-tile maximum), hard-negative candidates re-scored in fp32 on the
    owning rank; the shards' shadows are the slices of the whole pool's."""
    from vlsfr_amd.head import ShardedDcpHead
    Q, D, B, T, world = 32768, 512, 64, 2, 8
    case = common.head_case(4200, Q, D, B, T, n_id)
    full = make_head(case["queue0"], loss_type, 32.0, margin, False)
    full.head_dtype = "fp8"
    q0 = torch.from_numpy(case["queue0"]).cuda()
    Qs = Q // world
    shards = [ShardedDcpHead(q0[:, r * Qs:(r + 1) * Qs].contiguous(), r, world, Q, 32.0, margin, loss_type, precise=False)
              for r in range(world)]
    for h in shards:
        h.head_dtype = "fp8"
    for t in range(T):
        xl, yl = case["XL"][t], case["YL"][t]
        for s_, (pl, gl, trans) in enumerate(((xl, yl, True), (yl, xl, False))):
            g = torch.from_numpy(case["G"][t, s_]).cuda()
            p = torch.from_numpy(case["P"][t, s_]).cuda().requires_grad_(True)
            loss = full.run_pass(p, g, pl, gl, trans)
            loss.backward()
            for l, dP in _sharded_pass(shards, p.detach(), g, pl, gl, trans):
                np.testing.assert_allclose(float(l), float(loss.detach()), rtol=1e-4, atol=1e-5)
                np.testing.assert_allclose(dP.cpu().numpy(), p.grad.cpu().numpy(), rtol=2e-3,
                                           atol=2e-4 * float(p.grad.abs().max()))
    assert all(h.shadow.t.get("fp8") is not None and h.shadow.t.get("bf16") is None for h in shards)
    assert torch.equal(torch.cat([h.queue for h in shards], dim=1), full.queue)
    assert torch.equal(torch.cat([h.shadow.t["fp8"] for h in shards], dim=0), full.shadow.t["fp8"])
    for h in shards:
        assert h.lru.state_dict() == full.lru.state_dict() and h.qp.tolist() == full.qp.tolist()


# ---- config C4 (BASELINE configs[3]): 100 M identities, pool sharded by identity over 8 GPUs ------------------------------
class _LoneRankComm(object):
    """The collectives of ShardedDcpHead.combine as rank 0 of `world` sees them when every OTHER rank contributes an empty
    state (no columns: M = -inf, L = 0, O = T = 0, no candidates): the result is the loss / dL/dp of rank 0's own rows over
    rank 0's columns only, which a dense evaluation of the shard can check."""

    def __init__(self, world):
        self.world = world

    def all_reduce_max(self, t):
        return t

    def all_gather(self, t):
        fill = -1 if t.dtype == torch.int32 else -1e30
        return torch.cat([t.unsqueeze(0), torch.full((self.world - 1,) + tuple(t.shape), fill, dtype=t.dtype, device=t.device)])

    def reduce_scatter_rows(self, t):
        return t[:t.shape[0] // self.world].clone()


def test_c4_rank0_shard_of_a_100m_identity_pool():
    """BASELINE configs[3] / SURVEY 8(d) C4 on ONE GPU: rank 0's shard of a 104 857 600-identity pool split over 8 ranks
    (13 107 200 slots: 53.7 GB fp32 master + 13.4 GB shadow), the replicated 100 M-entry LRU restored from arrays, the
    gathered batch of 8 x 64 rows.  The oracle cannot run at this size; checked instead, for the bf16 and the e4m3 sweep:
    (1) the transactional pass leaves pool, shadow, LRU and queue_position bit-identical; (2) the partial softmax state does
    not depend on the column partition (n_chunks); (3) rank 0's loss share and dL/dp rows over its 13.1 M columns equal a
    dense fp32 PyTorch evaluation of the shard (target terms only for the labels rank 0 owns); (4) the committing pass
    writes exactly the owned slots (fp32 master rows = g bit for bit, shadow = their cast), ffc.py:29-30,153-260."""
    import time
    from vlsfr_amd.ffc import build_pool
    from vlsfr_amd.head import ShardedDcpHead
    from vlsfr_amd.lru import LRU
    world, D, Bg = 8, 512, 512
    Q = 100 << 20
    Qs = Q // world
    t0 = time.perf_counter()
    shard = build_pool(Q, D, "cuda", shard=(0, world), seed=7)
    torch.cuda.synchronize()
    t_pool = time.perf_counter() - t0
    assert shard.shape == (2, Qs, D)
    t0 = time.perf_counter()
    lru = LRU(Q)
    ar = np.arange(Q, dtype=np.int64)
    lru.restore_arrays(ar, ar.astype(np.int32))            # steady state: identity k resident in slot k (lru.py:113)
    del ar
    t_lru = time.perf_counter() - t0
    qp = np.zeros(Q, dtype=np.uint8)
    rng = np.random.default_rng(4)
    # labels of the gathered batch (main.py:53-60 per rank: id half shared between the views, instance half not); a
    # quarter of them are identities rank 0 owns, the rest live on the other ranks
    own = rng.choice(Qs, size=Bg // 2, replace=False)
    other = Qs + rng.choice(Q - Qs, size=3 * Bg, replace=False)
    xl = np.concatenate([own[:Bg // 4], other[:3 * Bg // 4]]).astype(np.int64)
    yl = np.concatenate([own[:Bg // 8], own[Bg // 4:3 * Bg // 8], other[3 * Bg // 4:3 * Bg // 2]]).astype(np.int64)
    perm = rng.permutation(Bg)
    xl, yl = xl[perm], yl[perm]
    p_np, g_np = common.unit_rows(rng, Bg, D), common.unit_rows(rng, Bg, D)
    p_all, g_all = torch.from_numpy(p_np).cuda(), torch.from_numpy(g_np).cuda()
    before_q = shard.clone()
    before_lru = lru.state_arrays()
    comm = _LoneRankComm(world)
    B = Bg // world
    for dtype in ("bf16", "fp8"):
        head = ShardedDcpHead(shard, 0, world, Q, 32.0, 0.5, "Arc", precise=False, lru=lru, qp=qp)
        head.head_dtype = dtype
        assert head.hard_neg == 10
        outs = []
        for n_chunks in (0, 512):
            head.n_chunks = n_chunks
            head.finish(head.combine(head.partial(p_all, g_all, xl, yl, True), comm, own_rows=(0, B)))   # builds the shadow, warms the kernels
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            st = head.partial(p_all, g_all, xl, yl, True)
            st = head.combine(st, comm, own_rows=(0, B))
            loss, dP = head.finish(st)
            e1.record()
            torch.cuda.synchronize()
            outs.append((float(loss), dP.cpu().numpy(), st["M"].cpu().numpy(), head.unpack(st)["L"].cpu().numpy()))
            assert st["fixed_ref"]                           # the shadow sweeps need no all-reduce(max)
            print("C4 rank-0 shard, %s sweep, n_chunks %d: %.2f ms per pass (%d gathered rows x %d slots)" %
                  (dtype, n_chunks, e0.elapsed_time(e1), Bg, Qs))
        # (1) rollback: nothing moved
        assert torch.equal(shard, before_q)
        assert _shadow_matches_pool(head, dtype, shard)
        after = lru.state_arrays()
        assert np.array_equal(before_lru[0], after[0]) and np.array_equal(before_lru[1], after[1])
        assert not qp.any()
        # (2) column partition
        (l0, d0, M0, L0), (l1, d1, M1, L1) = outs
        assert np.isfinite(l0) and np.isfinite(d0).all()
        np.testing.assert_allclose(l0, l1, rtol=1e-5 if dtype == "bf16" else 1e-4)
        np.testing.assert_allclose(d0, d1, rtol=2e-3, atol=2e-4 * np.abs(d0).max())
        np.testing.assert_allclose(M0 + np.log2(L0), M1 + np.log2(L1), rtol=1e-5 if dtype == "bf16" else 1e-4)
        # (3) dense fp32 evaluation of rank 0's own rows over its columns
        ysl = torch.from_numpy(yl).cuda()
        tgt = torch.from_numpy(xl[:B]).cuda()               # full residency: identity k sits in slot k
        lab_slot = torch.where(tgt < Qs, tgt, torch.full_like(tgt, -1))
        loss_ref, dp_ref = _dense_shard_reference(p_all[:B], shard, g_all, ysl, lab_slot, Bg)
        np.testing.assert_allclose(l0, loss_ref, rtol=2e-3 if dtype == "bf16" else 5e-3)
        num = (d0 * dp_ref).sum(1)
        den = np.linalg.norm(d0, axis=1) * np.linalg.norm(dp_ref, axis=1)
        assert (num / den).min() > (0.999 if dtype == "bf16" else 0.99)
        assert np.linalg.norm(d0 - dp_ref) <= (2e-2 if dtype == "bf16" else 8e-2) * np.linalg.norm(dp_ref)
    # (4) the committing pass (last: it changes pool and allocator state)
    head.n_chunks = 0
    st = head.partial(p_all, g_all, xl, yl, False)
    st = head.combine(st, comm, own_rows=(0, B))
    head.finish(st)
    torch.cuda.synchronize()
    want = before_q
    wr_np = yl < Qs
    want[0, torch.from_numpy(yl[wr_np]).cuda()] = g_all[torch.from_numpy(wr_np).cuda()]     # hits with queue_position 0 -> row 0
    assert torch.equal(shard, want)
    assert _shadow_matches_pool(head, "fp8", shard)
    assert qp[yl].tolist() == [1] * Bg and int(qp.sum()) == Bg        # replicated state: every rank flips all of them
    keys, slots = lru.state_arrays()
    assert keys[:Bg].tolist() == yl[::-1].tolist()                      # most recent first (lru.py:44-89)
    lru_bytes = 4 * 2 * (Q + 2) + 8 * Q + 4 * (1 << 28) + Q
    print("C4: shard build %.1f s, 100 M-entry LRU restore %.1f s; host bytes per replicated LRU + queue_position: %.2f GB; "
          "HBM: pool %.1f GB + shadow %.1f GB" % (t_pool, t_lru, lru_bytes / 1e9, shard.numel() * 4 / 1e9, Qs * 1024 / 1e9))


def _dense_shard_reference(p, shard, g_all, gallery_slot, label_slot, n_pos):
    """Dense fp32 loss share and dL/dp of the rows `p` over one pool shard (Arc, margin 0.5, scale 32): both variants of
    ffc.py:195-201 — variant 1 reads g at the slots the gallery batch writes into this shard, variant 2 the other pool row
    there (every label hits with queue_position 0: ones_idx = all written slots) — the margin on the target column if this
    shard owns it (label_slot >= 0), softmax over this shard's columns only, loss normalised by the batch's n_pos."""
    B, Qs = p.shape[0], shard.shape[1]
    wr = (gallery_slot >= 0) & (gallery_slot < Qs)
    cols = gallery_slot[wr]
    tot, dp = 0.0, torch.zeros_like(p)
    own = label_slot >= 0
    cm, sm = float(np.cos(0.5)), float(np.sin(0.5))
    for v in range(2):
        wv = g_all[wr] if v == 0 else shard[1, cols]
        cos = torch.empty(B, Qs, device=p.device)
        for a0 in range(0, Qs, 1 << 21):
            cos[:, a0:a0 + (1 << 21)] = p @ shard[0, a0:a0 + (1 << 21)].t()
        cos[:, cols] = p @ wv.t()
        fac = torch.ones(B, device=p.device)
        ls = label_slot.clamp(min=0).view(-1, 1)
        gt = cos.gather(1, ls).view(-1)
        tm = gt * cm - torch.sqrt(1 - gt * gt) * sm
        dtm = cm + gt / torch.sqrt(1 - gt * gt) * sm
        z = cos * 32.0
        z.scatter_(1, ls, torch.where(own, tm * 32.0, z.gather(1, ls).view(-1)).view(-1, 1))
        lse = torch.logsumexp(z, dim=1)
        zt = torch.where(own, tm * 32.0, torch.zeros_like(tm))
        tot += float((lse - zt).sum()) / n_pos
        sm_w = torch.softmax(z, dim=1)                                  # [B, Qs]
        sm_w.scatter_(1, ls, torch.where(own, sm_w.gather(1, ls).view(-1) * dtm, sm_w.gather(1, ls).view(-1)).view(-1, 1))
        # dL/dp = scale / n_pos * (sum_j softmax_j * dz_j/dcos_j * w_j  -  [owned] dtm * w_target)
        wsp = sm_w[:, cols].clone()
        sm_w[:, cols] = 0
        acc = torch.zeros_like(p)
        for a0 in range(0, Qs, 1 << 21):
            acc += sm_w[:, a0:a0 + (1 << 21)] @ shard[0, a0:a0 + (1 << 21)]
        acc += wsp @ wv
        # the target's class vector: the written value if the gallery batch writes that slot, else pool row 0
        hit = (ls == cols.view(1, -1))
        has = hit.any(1)
        wt = torch.where(has.view(-1, 1), wv[hit.float().argmax(1)], shard[0, ls.view(-1)])
        acc -= torch.where(own, dtm, torch.zeros_like(dtm)).view(-1, 1) * wt
        dp += 32.0 / n_pos * acc
        del cos, z, sm_w
    return tot, dp.cpu().numpy()
