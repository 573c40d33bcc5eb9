"""Native (C++, libvlsfr.so) LRU + DCP bookkeeping vs the reference's golden traces and vs the
oracle under random op sequences.  Host-only: runs without a GPU."""
import ctypes

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle.ffc_ref import dcp_assign_ref
from oracle.lru_ref import LRURef
from tests.test_oracle_golden import replay_lru_trace
from vlsfr_amd import _lib
from vlsfr_amd.lru import LRU


def test_native_lru_matches_reference_traces():
    replay_lru_trace(LRU, lambda l: [o.op_type for o in l.op_stack])


def test_known_answers_capacity3():
    """SURVEY.md §8 (a7-detail), measured on the reference."""
    l = LRU(3)
    assert [l.get(k) for k in (10, 20, 30)] == [0, 1, 2]
    assert l.state_dict() == [(30, 2), (20, 1), (10, 0)]
    assert l.get(10) == 0 and l.state_dict() == [(10, 0), (30, 2), (20, 1)]
    assert l.get(40) == 1 and l.view(20) == -1 and l.view(40) == 1
    assert [l.try_get(k) for k in (50, 10, 60)] == [2, 0, 1]
    assert [o.op_type for o in l.op_stack] == ["Overflow", "Get", "Overflow"]
    l.rollback_steps(3)
    assert l.state_dict() == [(40, 1), (10, 0), (30, 2)] and l.cur_idx == 3 and len(l.op_stack) == 0
    l2 = LRU(4)
    assert l2.try_get(1) == 0 and l2.op_stack[0].op_type == "Add"
    l2.rollback_one_step()
    assert l2.cur_idx == 0 and 1 not in l2 and len(l2.cache) == 0
    l2.rollback_steps(5)          # clamped to the stack depth (lru.py:252-255)


def test_api_surface_and_errors():
    l = LRU(2)
    assert l.capacity == 2 and l.cur_idx == 0 and len(l.cache) == 0 and list(l) == []
    l.get(5)
    assert 5 in l and 6 not in l and "x" not in l and list(l.keys()) == [5] and l.cache[5] == 0
    with pytest.raises(TypeError):
        l.get("a")
    with pytest.raises(AssertionError):
        l.restore([(1, 0)])            # not empty
    l3 = LRU(2)
    with pytest.raises(AssertionError):
        l3.restore([(1, 0), (2, 1), (3, 0)])   # more than capacity
    with pytest.raises(AssertionError):
        l3.restore([(1, 0), (1, 1)])   # duplicate key
    with pytest.raises(_lib.VlsfrError):
        LRU(0)


ops_strategy = st.lists(st.tuples(st.sampled_from(["get", "try_get", "view", "rollback", "contains"]),
                                  st.integers(0, 11)), min_size=1, max_size=120)


@settings(max_examples=150, deadline=None)
@given(cap=st.integers(1, 7), ops=ops_strategy)
def test_native_lru_equals_oracle(cap, ops):
    a, b = LRU(cap), LRURef(cap)
    for kind, k in ops:
        if kind == "get":
            if len(b.undo):      # committing gets only on an empty undo stack (the way ffc.py uses it)
                kind = "try_get"
            else:
                assert a.get(k) == b.get(k)
        if kind == "try_get":
            assert a.try_get(k) == b.try_get(k)
        elif kind == "view":
            assert a.view(k) == b.view(k)
        elif kind == "contains":
            assert (k in a) == (k in b)
        elif kind == "rollback":
            a.rollback_steps(k)
            b.rollback_steps(k)
        assert a.state_dict() == b.state_dict()
        assert a.cur_idx == b.cur_idx
        assert [o.op_type for o in a.op_stack] == b.op_types()
        assert len(a.cache) == len(b.slot)


def native_assign(lru, qp, gl, pl, transactional):
    n = len(gl)
    L = _lib.lib()
    gl_a, pl_a = np.asarray(gl, dtype=np.int64), np.asarray(pl, dtype=np.int64)
    i32 = lambda m: np.zeros(max(m, 1), dtype=np.int32)
    rows, cols, lab, ones = i32(n), i32(n), i32(n), i32(n)
    sc, s1, s2, us = i32(3 * n), i32(3 * n), i32(3 * n), i32(n)
    uv = np.zeros(max(n, 1), dtype=np.uint8)
    plan = _lib.DcpPlan()
    p = lambda a: a.ctypes.data
    _lib.check(L.vlsfr_dcp_assign(lru._h, p(qp), p(gl_a), p(pl_a), n, int(transactional), p(rows), p(cols), p(lab),
                                  p(ones), p(sc), p(s1), p(s2), p(us), p(uv), ctypes.byref(plan)))
    return dict(rows=rows[:n].tolist(), cols=cols[:n].tolist(), labels=lab[:n].tolist(),
                ones=ones[:plan.n_ones].tolist(), special=sc[:plan.n_special].tolist(),
                src1=s1[:plan.n_special].tolist(), src2=s2[:plan.n_special].tolist(), plan=plan,
                undo=(us, uv))


@settings(max_examples=120, deadline=None)
@given(cap=st.integers(2, 9), batches=st.lists(st.tuples(st.lists(st.integers(0, 14), min_size=1, max_size=8),
                                                         st.booleans()), min_size=1, max_size=8), data=st.data())
def test_dcp_assign_equals_oracle(cap, batches, data):
    a, b = LRU(cap), LRURef(cap)
    qa, qb = np.zeros(cap, dtype=np.uint8), [0] * cap
    L = _lib.lib()
    for gl, transactional in batches:
        pl = data.draw(st.lists(st.integers(0, 14), min_size=len(gl), max_size=len(gl)))
        got = native_assign(a, qa, gl, pl, transactional)
        rows, cols, labels, ones, saved = dcp_assign_ref(b, qb, gl, pl, transactional)
        assert (got["rows"], got["cols"], got["labels"], got["ones"]) == (rows, cols, labels, ones)
        assert got["plan"].n_pos == sum(1 for v in labels if v >= 0)
        # special columns = written slots ∪ ones ∪ positive labels, each once
        want = set(cols) | set(ones) | set(v for v in labels if v >= 0)
        assert set(got["special"]) == want and len(got["special"]) == len(want)
        last = {}
        for i, (r, c) in enumerate(zip(rows, cols)):
            last[(r, c)] = i
        for c, s1, s2 in zip(got["special"], got["src1"], got["src2"]):
            assert s1 == last.get((0, c), -1)
            assert s2 == (last.get((1, c), -2) if c in ones else s1)
        if transactional:
            us, uv = got["undo"]
            _lib.check(L.vlsfr_dcp_undo(a._h, qa.ctypes.data, us.ctypes.data, uv.ctypes.data, ctypes.byref(got["plan"])))
            for k, v in saved.items():
                qb[k] = v
            b.rollback_steps(len(gl))
        assert qa.tolist() == qb
        assert a.state_dict() == b.state_dict() and a.cur_idx == b.cur_idx and len(a.op_stack) == len(b.undo)


def test_lru_large_capacity_roundtrip():
    """1M-slot pool (config C2): fill, evict, state/restore round trip, rollback restores exactly."""
    Q = 1 << 20
    l = LRU(Q)
    rng = np.random.default_rng(0)
    keys = rng.permutation(3 * Q)[: Q + 1000].astype(np.int64)
    L = _lib.lib()
    out = ctypes.c_int32()
    for k in keys.tolist()[:5]:
        l.get(k)
    qp = np.zeros(Q, dtype=np.uint8)
    B = 4096
    for s in range(5, len(keys), B):
        chunk = keys[s:s + B]
        native_assign(l, qp, chunk.tolist(), chunk.tolist(), False)
    assert l.cur_idx == Q and len(l.cache) == Q
    before = l._state_arrays()
    got = native_assign(l, qp.copy(), keys[:512].tolist(), keys[:512].tolist(), True)
    l.rollback_steps(512)
    after = l._state_arrays()
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    l2 = LRU(Q)
    l2.restore(zip(before[0].tolist(), before[1].tolist()))
    a2 = l2._state_arrays()
    assert np.array_equal(before[0], a2[0]) and np.array_equal(before[1], a2[1])
