"""reranker blend + result assembly (rag_engine.rs:599-700).  Pure host
arithmetic in csrc/engine.cpp, so it is checked against the oracle without a GPU."""
import ctypes as C
import importlib

import numpy as np

from conftest import bits


def _blend(rlr, cand_rows, cand_init, rer_rows, rer_rel, top_k, weights=None):
    N = importlib.import_module("rust-local-rag_amd._native")
    n = len(cand_rows)
    cand = (N.SearchHitC * max(n, 1))()
    for i in range(n):
        cand[i].row = int(cand_rows[i])
        cand[i].score = cand[i].initial_score = float(cand_init[i])
    rr = np.ascontiguousarray(rer_rows if len(rer_rows) else [0], dtype=np.uint64)
    rs = np.ascontiguousarray(rer_rel if len(rer_rel) else [0], dtype=np.float32)
    out = (N.SearchHitC * max(n, 1))()
    rer = np.zeros(max(n, 1), np.float32)
    has = np.zeros(max(n, 1), np.int32)
    n_out = C.c_uint32()
    wc = weights.to_c() if weights is not None else None
    st = N.lib().rlr_engine_blend_reranked(cand, n, rr.ctypes.data_as(N.u64p), rs.ctypes.data_as(N.f32p), len(rer_rows),
                                           top_k, C.byref(wc) if wc is not None else None, out,
                                           rer.ctypes.data_as(N.f32p), has.ctypes.data_as(N.i32p), max(n, 1), C.byref(n_out))
    assert st == 0
    k = n_out.value
    return ([int(out[i].row) for i in range(k)], np.array([out[i].score for i in range(k)], np.float32), rer[:k],
            has[:k].astype(bool))


def test_blend_hand_computed(rlr):
    # candidates by initial score; reranker likes row 30 most
    rows, init = [10, 20, 30, 40], [0.8, 0.6, 0.4, 0.2]
    got_rows, score, rer, has = _blend(rlr, rows, init, [30, 10, 99, 10], [0.9, 0.45, 0.7, 0.1], 3)
    # row 99 is not a candidate, the second 10 is a repeat -> both skipped (:617-618)
    f = np.float32
    b30 = f(0.7) * (f(0.9) / f(0.9)) + f(0.3) * (f(0.4) / f(0.8))
    b10 = f(0.7) * (f(0.45) / f(0.9)) + f(0.3) * (f(0.8) / f(0.8))
    assert got_rows == [30, 10, 20]                       # two blended, then the best unseen by initial score
    assert np.array_equal(bits(score), bits([b30, b10, f(0.6)]))
    assert list(has) == [True, True, False] and rer[0] == f(0.9) and rer[1] == f(0.45)


def test_blend_without_reranker_is_the_fallback_order(rlr):
    rows, init = [5, 6, 7], [0.1, 0.9, 0.5]
    got_rows, score, _, has = _blend(rlr, rows, init, [], [], 2)
    assert got_rows == [6, 7] and not has.any() and np.array_equal(bits(score), bits([0.9, 0.5]))


def test_blend_matches_oracle_on_random_cases(rlr, oracle):
    rng = np.random.default_rng(11)
    for trial in range(200):
        n = int(rng.integers(1, 40))
        rows = rng.permutation(1000)[:n]
        init = np.sort(rng.random(n).astype(np.float32))[::-1].copy()
        if trial % 7 == 0:
            init[rng.integers(0, n)] = init[0]            # ties
        m = int(rng.integers(0, n + 3))
        rer_rows = rng.choice(np.concatenate([rows, [2000, 2001]]), size=m, replace=True) if m else np.array([], np.int64)
        rer_rel = rng.random(m).astype(np.float32)
        if trial % 5 == 0 and m:
            rer_rel[:] = 0.0                              # max clamps to EPSILON
        top_k = int(rng.integers(1, n + 2))
        w = rlr.QueryWeights(reranker=float(np.float32(rng.random())), initial=float(np.float32(rng.random())))
        got_rows, score, rer, has = _blend(rlr, rows, init, rer_rows, rer_rel, top_k, w)
        oc, os_, orr, oh = oracle.blend(rows, init, rer_rows, rer_rel, top_k, w.reranker, w.initial)
        assert got_rows == [int(rows[i]) for i in oc], trial
        assert np.array_equal(bits(score), bits(os_)) and np.array_equal(has, oh)
        assert np.array_equal(bits(rer[has]), bits(orr[oh]))
