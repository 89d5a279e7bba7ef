"""The oracle against every golden vector the reference's own tests hold for the hot
path (tests/golden/reference_kats.json <- rag_engine.rs:2674-3226).  CPU only."""
import math

import numpy as np

from conftest import pf


def test_build_flags(oracle):
    assert b"ffp-contract=off" in oracle.lib().rlr_o_build_flags()


def test_cosine_kats(oracle, kats):
    for case in kats["cosine"]:
        got = oracle.cosine(case["a"], case["b"])
        if "exact" in case:
            assert got == case["exact"], case["name"]
        elif "approx" in case:
            assert abs(got - case["approx"]) < case["tol"], case["name"]
        else:
            lo, hi = case["range"]
            assert lo <= got <= hi, case["name"]


def test_cosine_ramp_and_normalized_dot(oracle, kats):
    c = kats["cosine_ramp"]
    dim = c["dim"]
    a = np.arange(dim, dtype=np.float32) / np.float32(dim)
    b = (np.arange(dim, dtype=np.float32) + np.float32(10)) / np.float32(dim)
    cos = oracle.cosine(a, b)
    lo, hi = c["cosine_open_range"]
    assert lo < cos < hi
    d = oracle.dot(oracle.normalize(a), oracle.normalize(b))
    assert abs(cos - d) < c["dot_of_normalized_equals_cosine_tol"]


def _run_mmr(oracle, case):
    cands = case["candidates"]
    if not cands:
        return []
    ids = [c[0] for c in cands]
    scores = np.array([pf(c[1]) for c in cands], dtype=np.float32)
    emb = np.array([c[2] for c in cands], dtype=np.float32)
    order, _ = oracle.mmr(emb, scores, case["top_k"], case["lambda"])
    return [ids[i] for i in order]


def check_mmr_case(case, got):
    if "expect_ids" in case:
        assert got == case["expect_ids"], case["name"]
    if "expect_len" in case:
        assert len(got) == case["expect_len"], case["name"]
    if "expect_first" in case:
        assert got[0] == case["expect_first"], case["name"]
    if "expect_absent" in case:
        assert case["expect_absent"] not in got, case["name"]


def test_mmr_kats(oracle, kats):
    for case in kats["mmr"]:
        check_mmr_case(case, _run_mmr(oracle, case))


def test_mmr_top_k_zero_still_returns_first(oracle):
    # `selected.push(remaining.swap_remove(0))` is unconditional (rag_engine.rs:782-785)
    order, _ = oracle.mmr(np.eye(3, dtype=np.float32), [0.9, 0.8, 0.7], 0, 0.3)
    assert list(order) == [0]


def test_mmr_swap_remove_visiting_order(oracle):
    # equal relevance, orthogonal embeddings: every MMR value ties, so the pick order is
    # the swap_remove-perturbed visiting order: [0], then rem=[3,1,2] -> 3, rem=[2,1] -> 2, 1
    order, _ = oracle.mmr(np.eye(4, dtype=np.float32), [0.5, 0.5, 0.5, 0.5], 4, 0.3)
    assert list(order) == [0, 3, 2, 1]


def test_resolve_weight_kats(oracle, kats):
    for case in kats["resolve_weight"]:
        ov = None if case["override"] is None else pf(case["override"])
        got = oracle.resolve_weight(ov, case["default"])
        exp = np.float32(pf(case["expect"]))
        assert np.float32(got) == exp, case
        if exp == 0:
            assert math.copysign(1.0, got) == math.copysign(1.0, float(exp))


def test_search_small_matches_bruteforce_definition(oracle):
    rows = oracle.synth_rows(257, 48, seed=7)
    q = oracle.synth_query(48, seed=8)
    r, c, e, l = oracle.search(rows, q, top_k=5)
    qn = oracle.normalize(q)
    e_all = np.array([oracle.dot(qn, rows[i]) for i in range(rows.shape[0])], dtype=np.float32)
    comb = np.float32(0.7) * e_all
    order = sorted(range(len(comb)), key=lambda i: (-float(comb[i]), i))[:5]
    assert list(r) == order
    assert np.array_equal(c, comb[order]) and np.array_equal(e, e_all[order])
    assert not l.any()
    # stage 1 = the 3*top_k reranker candidates (rag_engine.rs:544)
    r1 = oracle.search(rows, q, top_k=5, stage=1)[0]
    assert len(r1) == 15 and list(r1[:5]) == order
    # top_k = 0 is treated as 1 (rag_engine.rs:490)
    assert len(oracle.search(rows, q, top_k=0)[0]) == 1


def test_search_hybrid_lexical_blend(oracle):
    rows = oracle.synth_rows(100, 256, seed=3)
    q = oracle.synth_query(256, seed=4)
    lex = [(17, 2.0), (42, 4.0)]
    r, c, e, l = oracle.search(rows, q, top_k=3, lex=lex)
    # row 42 has lexical 1.0 -> 0.7*e + 0.3 dominates
    assert r[0] == 42 and l[0] == np.float32(1.0)
    i17 = list(r).index(17)
    assert l[i17] == np.float32(0.5)
    assert c[i17] == np.float32(0.7) * e[i17] + np.float32(0.3) * np.float32(0.5)


def test_search_with_diversity_pool_and_lambda_zero(oracle):
    rows = oracle.synth_rows(200, 32, seed=5, n_clusters=6)
    q = oracle.synth_query(32, seed=6)
    assert np.array_equal(oracle.search_with_diversity(rows, q, 5, 0.0)[0], oracle.search(rows, q, 5)[0])
    got = oracle.search_with_diversity(rows, q, 5, 0.7)[0]
    pool = oracle.search(rows, q, 15)  # max(3k, k+10) = 15
    order, _ = oracle.mmr(rows[pool[0].astype(np.int64)], pool[1], 5, 0.7)
    assert list(got) == [int(pool[0][i]) for i in order]
    assert got[0] == pool[0][0]


def test_synth_generator_is_deterministic_and_unit_norm(oracle):
    a = oracle.synth_rows(16, 768, seed=11)
    b = oracle.synth_rows(8, 768, seed=11, row0=8)
    assert np.array_equal(a[8:], b)
    n = np.linalg.norm(a.astype(np.float64), axis=1)
    assert np.all(np.abs(n - 1) < 1e-6)
    assert not np.array_equal(a, oracle.synth_rows(16, 768, seed=12))


def test_f16_rounding_matches_numpy(oracle):
    x = (np.random.default_rng(0).standard_normal(20000) * 0.05).astype(np.float32)
    assert np.array_equal(oracle.f32_to_f16_bits(x), x.astype(np.float16).view(np.uint16))
    assert np.array_equal(oracle.round_f16(x), x.astype(np.float16).astype(np.float32))
