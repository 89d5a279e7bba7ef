"""Parity of the HIP path (through the C ABI) with the oracle -- the tests proper.
Bit-exact: identical top-k row sets in identical order, scores equal as bit patterns."""
import numpy as np
import pytest

from conftest import bits, pf
from test_oracle_kats import check_mmr_case

pytestmark = pytest.mark.gpu


def make_index(rlr, rows, dtype="f32", normalize=False):
    ix = rlr.GpuIndex(rows.shape[1], dtype)
    ix.upload(rows, normalize=normalize)
    return ix


def oracle_topk(oracle, rows, qn, k):
    e = oracle.scan(rows, qn)
    key = np.where(np.isnan(e), -np.inf, e)
    order = np.lexsort((np.arange(len(e)), -key.astype(np.float64)))[:k]
    return order.astype(np.uint64), e[order]


# ---------------------------------------------------------------- KATs through the ABI
def test_cosine_kats_on_gpu(rlr, oracle, kats):
    """dot of normalised == cosine (rag_engine.rs:2776-2799), evaluated by the GPU."""
    for case in kats["cosine"]:
        a, b = np.array(case["a"], np.float32), np.array(case["b"], np.float32)
        if a.size != b.size or a.size == 0:
            continue  # cosine_similarity's length/empty guards are host-only (dead code in production)
        ix = make_index(rlr, b.reshape(1, -1), normalize=True)
        got = float(ix.score_rows(rlr.normalize(a), [0])[0])
        want = oracle.dot(oracle.normalize(a), oracle.normalize(b))
        assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32), case["name"]
        if "approx" in case:
            assert abs(got - case["approx"]) < case["tol"]
        ix.close()
    dim = kats["cosine_ramp"]["dim"]
    a = np.arange(dim, dtype=np.float32) / np.float32(dim)
    b = (np.arange(dim, dtype=np.float32) + np.float32(10)) / np.float32(dim)
    ix = make_index(rlr, b.reshape(1, -1), normalize=True)
    rows, cos = ix.search_topk(rlr.normalize(a), 1)
    assert abs(float(cos[0, 0]) - oracle.cosine(a, b)) < 1e-6
    ix.close()


def test_mmr_kats_on_gpu(rlr, kats):
    for case in kats["mmr"]:
        cands = case["candidates"]
        if not cands:
            ix = rlr.GpuIndex(3)
            order, _ = ix.mmr_select([], [], case["top_k"], case["lambda"])
            assert len(order) == 0
            continue
        ids = [c[0] for c in cands]
        emb = np.array([c[2] for c in cands], np.float32)
        ix = make_index(rlr, emb)  # stored as given, like the test twin (no normalisation)
        order, _ = ix.mmr_select(np.arange(len(ids)), [pf(c[1]) for c in cands], case["top_k"], case["lambda"])
        check_mmr_case(case, [ids[i] for i in order])
        ix.close()


# ---------------------------------------------------------------- scan + top-k
@pytest.mark.parametrize("n,dim,k", [(1000, 768, 5), (4096, 768, 100), (3000, 1024, 300), (777, 384, 50),
                                     (513, 100, 17), (50, 768, 100), (20000, 768, 900)])
def test_search_topk_bit_exact(rlr, oracle, n, dim, k):
    rows = oracle.synth_rows(n, dim, seed=100 + n)
    qn = oracle.normalize(oracle.synth_query(dim, seed=200 + n))
    ix = make_index(rlr, rows)
    got_rows, got_cos = ix.search_topk(qn, k)
    want_rows, want_cos = oracle_topk(oracle, rows, qn, k)
    assert got_rows.shape[1] == min(k, n)
    assert np.array_equal(got_rows[0], want_rows)
    assert np.array_equal(bits(got_cos[0]), bits(want_cos))
    ix.close()


@pytest.mark.parametrize("dim,dtype", [(384, "f32"), (768, "f16"), (1536, "f32"), (1536, "f16"), (2048, "f32"), (2048, "f16"),
                                       (1280, "f32"), (64, "f32"), (128, "f32"), (192, "f32"), (320, "f32"), (640, "f32"),
                                       (896, "f32"), (1152, "f32"), (128, "f16"), (384, "f16"), (640, "f16"), (2304, "f16"),
                                       (100, "f32"), (1792, "f32")])
def test_scan_kernel_variants_bit_exact(rlr, oracle, dim, dtype):
    """every row-pitch class of the scan: 1 KiB multiples (fixed kernel, CH 1..8), 256/512-byte rows (packed
    kernel: several rows per wave-load group), anything else (generic); ragged row counts hit the tail packs"""
    for n, k in ((4099, 20), (130, 130), (3, 2)):
        rows = oracle.synth_rows(n, dim, seed=300 + dim + n, n_clusters=7, f16=(dtype == "f16"))
        ix = make_index(rlr, rows, dtype=dtype)
        for qi in range(2):
            qn = oracle.normalize(oracle.synth_query(dim, seed=400 + dim + qi))
            r, c = ix.search_topk(qn, k)
            wr, wc = oracle_topk(oracle, rows, qn, k)
            assert np.array_equal(r[0], wr), (dim, dtype, n)
            assert np.array_equal(bits(c[0]), bits(wc)), (dim, dtype, n)
        ix.close()


def test_synthetic_fill_matches_oracle_generator(rlr, oracle):
    for dim, dtype, ncl in ((768, "f32", 0), (1024, "f16", 0), (96, "f32", 5)):
        ix = rlr.GpuIndex(dim, dtype)
        ix.fill_synthetic(300, seed=42, row0=1000, n_clusters=ncl)
        got = ix.fetch_rows(np.arange(300))
        want = oracle.synth_rows(300, dim, seed=42, row0=1000, n_clusters=ncl, f16=(dtype == "f16"))
        assert np.array_equal(bits(got), bits(want)), (dim, dtype)
        ix.close()


def test_upload_normalize_on_device_bit_exact(rlr, oracle):
    rng = np.random.default_rng(5)
    raw = (rng.standard_normal((500, 768)) * 3).astype(np.float32)
    raw[7] = 0          # zero vector stays as it is (norm_sq <= 1e-20)
    raw[8] = 1e-12      # tiny vector too
    ix = make_index(rlr, raw, normalize=True)
    want = np.stack([oracle.normalize(r) for r in raw])
    assert np.array_equal(bits(ix.fetch_rows(np.arange(500))), bits(want))
    ix.close()


def test_batched_queries_equal_looped_single_queries(rlr, oracle):
    rows = oracle.synth_rows(5000, 768, seed=9)
    qs = np.stack([oracle.normalize(oracle.synth_query(768, seed=300 + i)) for i in range(7)])
    ix = make_index(rlr, rows)
    r, c = ix.search_topk(qs, 20)
    for i in range(7):
        wr, wc = oracle_topk(oracle, rows, qs[i], 20)
        assert np.array_equal(r[i], wr) and np.array_equal(bits(c[i]), bits(wc))
    ix.close()


def test_duplicate_chunks_overflow_the_guard_band(rlr, oracle):
    """Thousands of identical rows tie exactly at the top: the band holds > 4096
    candidates and the large-candidate path must still return row-ascending ties."""
    base = oracle.synth_rows(3000, 768, seed=21)
    q = oracle.synth_query(768, seed=22)
    dup = oracle.normalize(q + np.float32(0.01) * base[0])
    rows = np.concatenate([base, np.repeat(dup[None, :], 6000, axis=0)])
    ix = make_index(rlr, rows)
    qn = oracle.normalize(q)
    got_rows, got_cos = ix.search_topk(qn, 10)
    assert list(got_rows[0]) == list(range(3000, 3010))
    assert np.all(bits(got_cos[0]) == bits([oracle.dot(qn, dup)])[0])
    assert ix.profile_read().n_retries >= 1
    # the same flood when the scan nominates over the binary16 image (wider band), single query and three at once
    # (the large-candidate path then re-scans with the f32 kernel against the image-derived threshold)
    ix.enable_batch_image(True, single_query=True)
    q2 = oracle.normalize(oracle.synth_query(768, seed=23))
    for qs in (qn[None, :], np.stack([q2, qn, q2])):
        r, c = ix.search_topk(qs, 10)
        for i in range(len(qs)):
            wr, wc = oracle_topk(oracle, rows, qs[i], 10)
            assert np.array_equal(r[i], wr) and np.array_equal(bits(c[i]), bits(wc))
    ix.close()


def test_dense_scores_under_the_8bit_nomination_take_the_second_level(rlr, oracle):
    """a tight cluster around the query: tens of thousands of cosines inside the 8-bit nomination band, k in the hundreds to
    thousands -- the band outgrows the 4096-entry LDS sort and the large-candidate path finishes by a one-workgroup radix
    select over the exactly re-scored keys (it was a global bitonic network).  Bit-equal to the oracle; same for the f32
    nomination with k = 3000 (> 1024 and the band near the limit) and for several queries at once."""
    n, dim = 60000, 256
    rng = np.random.default_rng(77)
    q = oracle.synth_query(dim, seed=78)
    centre = oracle.normalize(q)
    rows = oracle.synth_rows(n, dim, seed=79)
    tight = centre[None, :] + np.float32(0.02) * rng.standard_normal((30000, dim)).astype(np.float32)
    rows[:30000] = tight / np.linalg.norm(tight, axis=1, keepdims=True).astype(np.float32)
    ix = make_index(rlr, rows)
    qn = oracle.normalize(q)
    ix.enable_batch_image(False, q8=True)
    for k in (100, 900, 3000):
        ix.profile_read(reset=True)
        r, c = ix.search_topk(qn, k)
        wr, wc = oracle_topk(oracle, rows, qn, k)
        assert np.array_equal(r[0], wr) and np.array_equal(bits(c[0]), bits(wc)), k
    assert ix.profile_read().n_retries >= 1                     # at least the last one overflowed the LDS sort
    q2 = oracle.normalize(centre + np.float32(0.05) * oracle.synth_query(dim, seed=80))
    r, c = ix.search_topk(np.stack([qn, q2, qn]), 2000)
    for i, qq in enumerate((qn, q2, qn)):
        wr, wc = oracle_topk(oracle, rows, qq, 2000)
        assert np.array_equal(r[i], wr) and np.array_equal(bits(c[i]), bits(wc)), i
    ix.enable_batch_image(False)                                 # f32 nomination
    r, c = ix.search_topk(qn, 3000)
    wr, wc = oracle_topk(oracle, rows, qn, 3000)
    assert np.array_equal(r[0], wr) and np.array_equal(bits(c[0]), bits(wc))
    ix.close()


def test_nan_and_zero_rows(rlr, oracle):
    rows = oracle.synth_rows(300, 768, seed=31)
    rows[5] = 0.0
    rows[9, 3] = np.nan
    qn = oracle.normalize(oracle.synth_query(768, seed=32))
    ix = make_index(rlr, rows)
    got_rows, got_cos = ix.search_topk(qn, 300)
    assert got_rows[0, -1] == 9 and np.isnan(got_cos[0, -1])       # NaN orders last
    wr, wc = oracle_topk(oracle, rows, qn, 300)
    assert np.array_equal(got_rows[0], wr)
    assert np.array_equal(bits(got_cos[0][:-1]), bits(wc[:-1]))
    ix.close()


def test_fp16_rows(rlr, oracle):
    rows = oracle.synth_rows(4000, 1024, seed=41, f16=True)  # rounded to fp16 after normalisation
    qn = oracle.normalize(oracle.synth_query(1024, seed=42))
    ix = rlr.GpuIndex(1024, "f16")
    ix.upload(oracle.synth_rows(4000, 1024, seed=41), normalize=False)  # the library rounds on store
    got_rows, got_cos = ix.search_topk(qn, 100)
    wr, wc = oracle_topk(oracle, rows, qn, 100)
    assert np.array_equal(got_rows[0], wr) and np.array_equal(bits(got_cos[0]), bits(wc))
    ix.close()


# ---------------------------------------------------------------- mutation
def test_append_and_delete_rows(rlr, oracle):
    rows = oracle.synth_rows(1200, 768, seed=51)
    ix = make_index(rlr, rows[:700])
    assert ix.append(rows[700:]) == 700 and len(ix) == 1200
    dead = np.array([0, 5, 5, 699, 700, 1199, 300], dtype=np.uint64)
    ix.delete_rows(dead)
    keep = np.setdiff1d(np.arange(1200), dead)
    assert len(ix) == len(keep)
    assert np.array_equal(bits(ix.fetch_rows(np.arange(len(keep)))), bits(rows[keep]))
    qn = oracle.normalize(oracle.synth_query(768, seed=52))
    got_rows, got_cos = ix.search_topk(qn, 25)
    wr, wc = oracle_topk(oracle, rows[keep], qn, 25)
    assert np.array_equal(got_rows[0], wr) and np.array_equal(bits(got_cos[0]), bits(wc))
    with pytest.raises(rlr.RlrError):
        ix.delete_rows([len(keep)])
    ix.close()


def test_empty_index_and_k_zero(rlr):
    ix = rlr.GpuIndex(768)
    r, c = ix.search_topk(np.zeros(768, np.float32), 5)
    assert r.shape == (1, 0)
    ix.upload(np.ones((3, 768), np.float32), normalize=True)
    r, c = ix.search_topk(np.zeros(768, np.float32), 0)
    assert r.shape == (1, 0)
    ix.close()


# ---------------------------------------------------------------- engine level
def build_engine(rlr, rows, dtype="f32"):
    eng = rlr.RagEngine(rows.shape[1], dtype)
    ids = []
    per_doc = 250
    for d0 in range(0, rows.shape[0], per_doc):
        part = rows[d0:d0 + per_doc]
        ids += eng.add_document(f"doc{d0 // per_doc}.pdf", [f"chunk {d0 + i}" for i in range(len(part))], part,
                                pages=[1 + i % 7 for i in range(len(part))])
    return eng, ids


def test_engine_search_matches_oracle_c1(rlr, oracle):
    """BASELINE config 1: 1k x 768, single query, top_k=5, diversity 0.0."""
    raw = (np.random.default_rng(61).standard_normal((1000, 768))).astype(np.float32)
    eng, ids = build_engine(rlr, raw)
    rows = np.stack([oracle.normalize(r) for r in raw])
    q = oracle.synth_query(768, seed=62)
    got = eng.search_with_diversity(q, 5, 0.0)
    wr, wc, we, wl = oracle.search(rows, q, 5)
    assert [g.row for g in got] == list(wr)
    assert np.array_equal(bits([g.score for g in got]), bits(wc))
    assert np.array_equal(bits([g.embedding_score for g in got]), bits(we))
    assert all(g.chunk_id == ids[g.row] and g.document == f"doc{g.row // 250}.pdf" for g in got)
    # stage 1: the 3*top_k reranker candidates
    cand = eng.search(q, 5, stage=1)
    assert [g.row for g in cand] == list(oracle.search(rows, q, 5, stage=1)[0])
    eng.close()


def test_engine_hybrid_lexical_and_weights(rlr, oracle):
    eng, ids = build_engine(rlr, oracle.synth_rows(3000, 768, seed=71))
    rows = eng.index.fetch_rows(np.arange(3000))  # add_document re-normalises (rag_engine.rs:358-359)
    q = oracle.synth_query(768, seed=72)
    lex_rows = [(10, 3.5), (2999, 7.0), (1500, 0.25), (77, 7.0)]
    lex = [(ids[r], s) for r, s in lex_rows]
    for w, (we_, wl_) in ((None, (0.7, 0.3)), (rlr.QueryWeights(embedding=0.2, lexical=0.9), (0.2, 0.9)),
                          (rlr.QueryWeights(embedding=0.0), (0.0, 0.3))):
        got = eng.search(q, 10, weights=w, lexical=lex)
        wr, wc, we, wl = oracle.search(rows, q, 10, w_e=we_, w_l=wl_, lex=lex_rows)
        assert [g.row for g in got] == list(wr), w
        assert np.array_equal(bits([g.score for g in got]), bits(wc))
        assert np.array_equal(bits([g.lexical_score for g in got]), bits(wl))
    eng.close()


@pytest.mark.parametrize("n,dim,k,lam,ncl", [(3000, 768, 5, 0.3, 0), (20000, 768, 100, 0.3, 40),
                                              (5000, 1024, 100, 0.7, 25), (40, 768, 100, 0.5, 3)])
def test_engine_search_with_diversity_matches_oracle(rlr, oracle, n, dim, k, lam, ncl):
    rows = oracle.synth_rows(n, dim, seed=81 + n, n_clusters=ncl)
    ix_rows = rows
    eng = rlr.RagEngine(dim)
    eng.add_document("all.pdf", [str(i) for i in range(n)], ix_rows)  # re-normalising unit rows is what the reference does too
    stored = eng.index.fetch_rows(np.arange(n))
    q = oracle.synth_query(dim, seed=82 + n)
    got = eng.search_with_diversity(q, k, lam)
    wr, wc, we, wl = oracle.search_with_diversity(stored, q, k, lam)
    assert [g.row for g in got] == list(wr)
    assert np.array_equal(bits([g.score for g in got]), bits(wc))
    eng.close()


def test_config2_search_with_diversity_at_100k_matches_oracle(rlr, oracle):
    """BASELINE config 2 at its stated size: 100 000 x 768 f32, single query, top_k = 100, MMR lambda = 0.3
    (pool 300 -> 100), through the engine ABI, rows / scores / cosines bit-equal to rlr_o_search_with_diversity."""
    n, dim, k, lam = 100_000, 768, 100, 0.3
    rows = oracle.synth_rows(n, dim, seed=0x5EED0002, n_clusters=200)
    eng = rlr.RagEngine(dim)
    eng.index.fill_synthetic(n, seed=0x5EED0002, n_clusters=200)
    eng._chunks = [rlr.DocumentChunk(str(i), "synthetic", "", i) for i in range(n)]
    assert np.array_equal(eng.index.fetch_rows(np.arange(0, n, 911)).view(np.uint32), rows[::911].view(np.uint32))
    for s in (1, 2):
        q = oracle.synth_query(dim, seed=0x5EED0002 + s)
        got = eng.search_with_diversity(q, k, lam)
        wr, wc, we, wl = oracle.search_with_diversity(rows, q, k, lam)
        assert len(got) == k and [g.row for g in got] == list(wr)
        assert np.array_equal(bits([g.score for g in got]), bits(wc))
        assert np.array_equal(bits([g.embedding_score for g in got]), bits(we))
        plain = eng.search(q, k)                                   # and the no-MMR ordering at the same size
        pr, pc, _, _ = oracle.search(rows, q, k)
        assert [g.row for g in plain] == list(pr) and np.array_equal(bits([g.score for g in plain]), bits(pc))
    eng.close()


def test_search_with_diversity_dense_band_goes_back_to_the_two_call_path(rlr, oracle):
    """The fused search -> pool -> MMR enqueue orders the re-scored candidates inside pool_prepare_kernel (no sort_emit
    launch) when they are at most 1024; a band that holds more -- here 1500 copies of the row closest to the query --
    reports status 1 and the host takes the two-call path: same answer as the oracle either way."""
    n, dim = 6000, 768
    rows = oracle.synth_rows(n, dim, seed=515, n_clusters=4)
    q = oracle.synth_query(dim, seed=516)
    best = int(np.argmax(oracle.scan(rows, oracle.normalize(q))))
    rows[100:1600] = rows[best]                                 # 1500 exact duplicates: one cosine, 1500 candidates in the band
    eng, _ = build_engine(rlr, rows)
    stored = eng.index.fetch_rows(np.arange(n))
    for k, lam in ((10, 0.3), (100, 0.7), (5, 0.0)):
        got = eng.search_with_diversity(q, k, lam)
        wr, wc, we, _ = oracle.search_with_diversity(stored, q, k, lam)
        assert [g.row for g in got] == list(wr), (k, lam)
        assert np.array_equal(bits([g.score for g in got]), bits(wc)) and np.array_equal(bits([g.embedding_score for g in got]), bits(we))
    eng.close()


def _search_diverse(rlr, ix, qn, pool, k, lam, w_e=0.7, w_l=0.3):
    import ctypes as C
    N = rlr._native
    kk = max(min(max(k, 1), pool), 1)
    rows = np.zeros(kk, np.uint64)
    cos = np.zeros(kk, np.float32)
    sc = np.zeros(kk, np.float32)
    n, fb = C.c_uint32(), C.c_int32()
    N.check(N.lib().rlr_search_diverse(ix.handle, qn.ctypes.data_as(N.f32p), pool, k, lam, w_e, w_l, -1.0,
                                       rows.ctypes.data_as(N.u64p), cos.ctypes.data_as(N.f32p), sc.ctypes.data_as(N.f32p),
                                       C.byref(n), C.byref(fb)))
    return rows[:n.value], cos[:n.value], sc[:n.value], fb.value


@pytest.mark.parametrize("n,dim,dtype,k,lam,ncl", [(20_000, 768, "f32", 100, 0.3, 40), (5_000, 1024, "f16", 100, 0.7, 25),
                                                    (40, 768, "f32", 100, 0.5, 3), (3_000, 384, "f32", 5, 1.0, 0),
                                                    (9, 768, "f32", 0, 0.4, 0)])
def test_search_diverse_is_the_two_call_path_on_the_device(rlr, oracle, n, dim, dtype, k, lam, ncl):
    """rlr_search_diverse (scan -> select -> re-score -> pool order -> gather -> Gram -> greedy, one synchronisation)
    against rlr_o_search_with_diversity and against rlr_search_topk + rlr_mmr_select on the same index."""
    rows = oracle.synth_rows(n, dim, seed=1200 + n, n_clusters=ncl, f16=(dtype == "f16"))
    ix = rlr.GpuIndex(dim, dtype)
    ix.upload(rows)
    pool = max(3 * k, k + 10)
    for s in range(3):
        q = oracle.synth_query(dim, seed=1300 + n + s)
        qn = oracle.normalize(q)
        r, c, sc, fb = _search_diverse(rlr, ix, qn, pool, k, lam)
        assert fb == 0
        wr, wc, we, _ = oracle.search_with_diversity(rows, q, k, lam)
        assert np.array_equal(r, wr) and np.array_equal(bits(sc), bits(wc)) and np.array_equal(bits(c), bits(we))
        pr, pc = ix.search_topk(qn, min(pool, n))                  # the two-call path, same index
        psc = (np.float32(0.7) * pc[0]).astype(np.float32)
        order, _ = ix.mmr_select(pr[0], psc, k, lam)
        assert np.array_equal(r, pr[0][order]) and np.array_equal(bits(sc), bits(psc[order]))
    # other weights
    q = oracle.synth_query(dim, seed=77)
    r, c, sc, fb = _search_diverse(rlr, ix, oracle.normalize(q), pool, k, lam, w_e=0.2, w_l=0.9)
    wr, wc, we, _ = oracle.search_with_diversity(rows, q, k, lam, w_e=0.2, w_l=0.9)
    assert fb == 0 and np.array_equal(r, wr) and np.array_equal(bits(sc), bits(wc))
    # arguments the fused kernels do not cover are handed back, not guessed at
    assert _search_diverse(rlr, ix, oracle.normalize(q), 2000, 100, lam)[3] != 0 or n <= 1024
    assert _search_diverse(rlr, ix, oracle.normalize(q), pool, k, lam, w_e=0.0)[3] != 0
    ix.close()


def test_engine_diversity_when_distinct_cosines_round_to_one_score(rlr, oracle):
    """embedding weight 1e-40: w_e * cos underflows to a handful of subnormal values, so long chains of DISTINCT
    cosines share one combined score and the candidate order (combined desc, row asc -- rag_engine.rs:543 with the
    build's tie rule) is no longer the cosine order.  The fused path must either order such chains itself or hand the
    query back when a chain reaches its last fetched row; either way the engine's answer equals the oracle's."""
    n, dim = 6000, 768
    rows = oracle.synth_rows(n, dim, seed=5150, n_clusters=9)
    eng = rlr.RagEngine(dim)
    eng.index.upload(rows)
    eng._chunks = [rlr.DocumentChunk(str(i), "synthetic", "", i) for i in range(n)]
    for w_e in (1e-40, 3e-39, 1e-3):
        w = rlr.QueryWeights(embedding=w_e)
        for k, lam in ((5, 0.3), (100, 0.5)):
            q = oracle.synth_query(dim, seed=5151 + k)
            got = eng.search_with_diversity(q, k, lam, weights=w)
            wr, wc, we, _ = oracle.search_with_diversity(rows, q, k, lam, w_e=w_e)
            assert [g.row for g in got] == list(wr), (w_e, k)
            assert np.array_equal(bits([g.score for g in got]), bits(wc)), (w_e, k)
            assert np.array_equal(bits([g.embedding_score for g in got]), bits(we)), (w_e, k)
    eng.close()


def test_mmr_select_values_bit_exact(rlr, oracle):
    rows = oracle.synth_rows(2000, 768, seed=91, n_clusters=12)
    ix = make_index(rlr, rows)
    qn = oracle.normalize(oracle.synth_query(768, seed=92))
    pr, pc = ix.search_topk(qn, 300)
    scores = (np.float32(0.7) * pc[0]).astype(np.float32)
    order, mmr = ix.mmr_select(pr[0], scores, 100, 0.3)
    worder, wmmr = oracle.mmr(rows[pr[0].astype(np.int64)], scores, 100, 0.3)
    assert np.array_equal(order, worder)
    assert np.isnan(mmr[0]) and np.array_equal(bits(mmr[1:]), bits(wmmr[1:]))
    # k = 0 still yields the first candidate
    assert list(ix.mmr_select(pr[0], scores, 0, 0.3)[0]) == [0]
    ix.close()


@pytest.mark.parametrize("P", [65, 129, 200, 256, 257, 283, 284, 300, 512, 513, 700, 768, 769, 1000, 1024])
def test_mmr_single_pool_sizes_and_awkward_relevance(rlr, oracle, P):
    """pool sizes around every per-lane slot count of the register-resident greedy kernel, relevance with ties / zeros of
    both signs / non-finite values, duplicated rows (similarity ties), every lambda regime -- bit-equal picks and logged MMR
    values, single call and batch entry point alike"""
    dim = 96
    rows = oracle.synth_rows(P + 50, dim, seed=7000 + P, n_clusters=6)
    rows[10:20] = rows[3]                                   # exact duplicates inside the pool
    ix = make_index(rlr, rows)
    rng = np.random.default_rng(P)
    pool = rng.permutation(P + 50)[:P].astype(np.uint64)
    for lam, kk in ((0.3, 100), (0.0, 40), (1.0, 25), (0.7, P)):
        sc = (rng.standard_normal(P) * 0.2).astype(np.float32)
        sc[rng.integers(0, P, 12)] = sc[0]                  # relevance ties
        sc[rng.integers(0, P, 3)] = 0.0
        sc[rng.integers(0, P, 3)] = -0.0
        if lam != 0.7:
            sc[rng.integers(1, P, 2)] = np.nan              # non-finite relevance is never picked ...
            sc[rng.integers(1, P, 1)] = np.inf
            sc[rng.integers(1, P, 1)] = -np.inf
        o, m = ix.mmr_select(pool, sc, kk, lam)
        wo, wm = oracle.mmr(rows[pool.astype(np.int64)], sc, kk, lam)
        assert np.array_equal(o, wo), (P, lam, kk)
        assert np.array_equal(bits(m[1:]), bits(wm[1:])), (P, lam, kk)
        ob, mb, nb = ix.mmr_select_batch(np.tile(pool, (8, 1)), np.tile(sc, (8, 1)), np.full(8, P, np.uint32), kk, lam)
        assert nb[5] == len(o) and np.array_equal(ob[5][: nb[5]], o) and np.array_equal(bits(mb[5][1: nb[5]]), bits(m[1:]))
    ix.close()


def test_search_documents_defaults_and_caps(rlr, oracle):
    rows = oracle.synth_rows(2000, 768, seed=95)
    eng, _ = build_engine(rlr, rows)
    stored = eng.index.fetch_rows(np.arange(2000))
    q = oracle.synth_query(768, seed=96)
    res = eng.search_documents(rlr.SearchRequest(q))           # top_k 5, diversity 0.3
    assert [r.row for r in res] == list(oracle.search_with_diversity(stored, q, 5, 0.3)[0])
    res = eng.search_documents(rlr.SearchRequest(q, top_k=5000, diversity_factor=7.0))  # cap 100, clamp 1.0
    assert [r.row for r in res] == list(oracle.search_with_diversity(stored, q, 100, 1.0)[0])
    text = rlr.format_search_results(res[:2])
    assert text.startswith("**1. [") and "(page " in text
    eng.close()


def test_concurrent_readers(rlr, oracle):
    import threading
    rows = oracle.synth_rows(8000, 768, seed=97)
    ix = make_index(rlr, rows)
    qs = [oracle.normalize(oracle.synth_query(768, seed=400 + i)) for i in range(8)]
    want = [oracle_topk(oracle, rows, q, 30) for q in qs]
    errs = []

    def work(i):
        try:
            for _ in range(5):
                r, c = ix.search_topk(qs[i], 30)
                assert np.array_equal(r[0], want[i][0]) and np.array_equal(bits(c[0]), bits(want[i][1]))
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    ix.close()


# ---------------------------------------------------------------- sharded path, one rank
def test_sharded_index_world1_device_results(rlr, oracle):
    """rlr_search_topk_device + the torch-side merge on the GPU (the N>1 collective itself is
    covered by the gloo test in test_sharded_cpu.py)."""
    import importlib
    import torch  # noqa: F401
    sharded = importlib.import_module("rust-local-rag_amd.sharded")
    n, dim, k = 6000, 768, 40
    sh = sharded.ShardedIndex(dim, n, "f32", device=0, rank=0, world=1)
    sh.fill_synthetic(seed=505)
    rows = oracle.synth_rows(n, dim, seed=505)
    qs = np.stack([oracle.normalize(oracle.synth_query(dim, seed=600 + i)) for i in range(3)])
    got_rows, got_cos = sh.search_topk(qs, k)
    for i in range(3):
        wr, wc = oracle_topk(oracle, rows, qs[i], k)
        assert np.array_equal(got_rows[i].astype(np.uint64), wr)
        assert np.array_equal(bits(got_cos[i]), bits(wc))
    # k larger than the shard: padded tail is dropped
    sh2 = sharded.ShardedIndex(dim, 30, "f32", device=0, rank=0, world=1)
    sh2.fill_synthetic(seed=506)
    r2, c2 = sh2.search_topk(qs[0], 100)
    assert r2.shape == (1, 30)
    wr, wc = oracle_topk(oracle, oracle.synth_rows(30, dim, seed=506), qs[0], 100)
    assert np.array_equal(r2[0].astype(np.uint64), wr) and np.array_equal(bits(c2[0]), bits(wc))


def test_sharded_diversity_world1_winner_exchange_path(rlr, oracle):
    """search_with_diversity over a ShardedIndex: plan -> rlr_fetch_rows_device -> permutation ->
    rlr_mmr_select_values (the all-to-all / all-gather between them is covered by the gloo test)."""
    import importlib
    import torch  # noqa: F401
    sharded = importlib.import_module("rust-local-rag_amd.sharded")
    n, dim, k = 5000, 1024, 100
    for dtype in ("f32", "f16"):
        sh = sharded.ShardedIndex(dim, n, dtype, device=0, rank=0, world=1)
        sh.fill_synthetic(seed=515, n_clusters=25)
        rows = oracle.synth_rows(n, dim, seed=515, n_clusters=25, f16=(dtype == "f16"))
        qs_raw = [oracle.synth_query(dim, seed=620 + i) for i in range(5)]
        qs = np.stack([oracle.normalize(q) for q in qs_raw])
        for lam in (0.7, 0.0):
            got = sh.search_with_diversity_batch(qs, k, lam)
            for i in range(len(qs)):
                wr, wc, we, wl = oracle.search_with_diversity(rows, qs_raw[i], k, lam)
                assert np.array_equal(got[i][0].astype(np.uint64), wr), (dtype, lam, i)
                assert np.array_equal(bits(got[i][1]), bits(wc)), (dtype, lam, i)
        sh.index.close()


def test_sharded_step_overflow_marker_forces_a_consistent_redo(rlr, oracle):
    """the asynchronous sharded step (begin -> merge -> end): a shard whose guard band overflows marks its slot,
    the merge reports the marker, the step is redone on the synchronous path -- results still exact"""
    import importlib
    import torch  # noqa: F401
    sharded = importlib.import_module("rust-local-rag_amd.sharded")
    base = oracle.synth_rows(3000, 768, seed=21)
    q = oracle.synth_query(768, seed=22)
    dup = oracle.normalize(q + np.float32(0.01) * base[0])
    rows = np.concatenate([base, np.repeat(dup[None, :], 6000, axis=0)])
    sh = sharded.ShardedIndex(768, len(rows), "f32", device=0, rank=0, world=1)
    sh.upload_shard(rows)
    qn = oracle.normalize(q)
    # the marker itself: begin/end + merge, without the redo
    local = torch.zeros((1, 10), dtype=torch.int64, device="cuda")
    t = sh.index.search_topk_device_begin(qn, 10, local.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert sh.index.search_topk_device_end(t) == 1
    torch.cuda.synchronize()
    assert int(local[0, 0].item()) == -1                       # all-ones word
    # the full step recovers
    r, c = sh.search_topk(qn, 10)
    assert r[0].tolist() == list(range(3000, 3010))
    assert np.all(bits(c[0]) == bits([oracle.dot(qn, dup)])[0])
    # and an ordinary query goes through the asynchronous path unchanged
    q2 = oracle.normalize(oracle.synth_query(768, seed=23) - q)
    r2, c2 = sh.search_topk(q2, 10)
    wr, wc = oracle_topk(oracle, rows, q2, 10)
    assert np.array_equal(r2[0].astype(np.uint64), wr) and np.array_equal(bits(c2[0]), bits(wc))
    sh.index.close()


def test_replicated_index_world1_and_query_sharding(rlr, oracle):
    """replicas-only mode: every rank holds the corpus, queries are dealt round-robin"""
    import importlib
    import torch  # noqa: F401
    sharded = importlib.import_module("rust-local-rag_amd.sharded")
    n, dim, k = 3000, 768, 15
    rows = oracle.synth_rows(n, dim, seed=707)
    qs = np.stack([oracle.normalize(oracle.synth_query(dim, seed=710 + i)) for i in range(5)])
    rep = sharded.ReplicatedIndex(dim, "f32", device=0, rank=0, world=1)
    rep.index.upload(rows)
    r, c = rep.search_topk(qs, k)
    for i in range(5):
        wr, wc = oracle_topk(oracle, rows, qs[i], k)
        assert np.array_equal(r[i].astype(np.uint64), wr) and np.array_equal(bits(c[i]), bits(wc))
    # rank 1 of 2 (no process group: gather is skipped) answers only its own queries
    rep1 = sharded.ReplicatedIndex(dim, "f32", device=0, rank=1, world=2, index=rep.index)
    assert rep1.my_queries(5).tolist() == [1, 3]
    r1, c1 = rep1.search_topk(qs, k, gather=False)
    assert np.array_equal(r1[[1, 3]], r[[1, 3]]) and (r1[[0, 2, 4]] == -1).all() and np.isnan(c1[0]).all()
    rep.index.close()


def test_mmr_select_values_entry_equals_index_pools(rlr, oracle):
    import torch
    rows = oracle.synth_rows(3000, 768, seed=93, n_clusters=9)
    ix = make_index(rlr, rows)
    qs = np.stack([oracle.normalize(oracle.synth_query(768, seed=94 + i)) for i in range(70)])  # > one 64-query pass
    pr, pc = ix.search_topk(qs, 40)
    scores = (np.float32(0.7) * pc).astype(np.float32)
    sizes = np.full(len(qs), 40, np.uint32)
    sizes[3] = 17
    want = ix.mmr_select_batch(pr, scores, sizes, 10, 0.6)
    vals = torch.empty((len(qs) * 40, 768), dtype=torch.float32, device="cuda")
    ix.fetch_rows_device(pr.ravel(), vals.data_ptr())
    assert np.array_equal(vals.cpu().numpy(), rows[pr.ravel().astype(np.int64)])
    got = ix.mmr_select_values(vals.data_ptr(), scores, sizes, 10, 0.6)
    assert np.array_equal(got[2], want[2])
    for q in range(len(qs)):
        nsel = got[2][q]
        assert np.array_equal(got[0][q, :nsel], want[0][q, :nsel])
        assert np.array_equal(bits(got[1][q, 1:nsel]), bits(want[1][q, 1:nsel]))
    ix.close()


# ---------------------------------------------------------------- batched (matrix-core) path
def _check_batch(rlr, oracle, ix, rows, qs, k):
    ix.profile_read(reset=True)
    r, c = ix.search_topk(qs, k)
    prof = ix.profile_read()
    for i in range(len(qs)):
        wr, wc = oracle_topk(oracle, rows, qs[i], k)
        assert np.array_equal(r[i], wr), f"query {i}: rows differ"
        assert np.array_equal(bits(c[i]), bits(wc)), f"query {i}: scores differ"
    return prof


def test_batched_mfma_small_corpus_all_materialised(rlr, oracle):
    rows = oracle.synth_rows(6000, 768, seed=111)
    qs = np.stack([oracle.normalize(oracle.synth_query(768, seed=700 + i)) for i in range(40)])
    ix = make_index(rlr, rows)
    prof = _check_batch(rlr, oracle, ix, rows, qs, 100)
    assert prof.n_batches == 1 and prof.n_batch_queries == 40 and prof.n_batch_fallbacks == 0
    prof = _check_batch(rlr, oracle, ix, rows, qs[:17], 5)
    assert prof.n_batches == 1 and prof.n_batch_fallbacks == 0
    ix.close()


def test_batched_mfma_sample_then_filter(rlr, oracle):
    n = 200_000
    rows = oracle.synth_rows(n, 768, seed=112)
    qs = np.stack([oracle.normalize(oracle.synth_query(768, seed=800 + i)) for i in range(24)])
    ix = rlr.GpuIndex(768)
    ix.fill_synthetic(n, seed=112)
    prof = _check_batch(rlr, oracle, ix, rows, qs, 100)
    assert prof.n_batches == 1 and prof.n_batch_fallbacks == 0
    # cost model at 200 k rows: two single scans are cheaper than any batched pipeline; four share one scan
    # (scan_multi_kernel); nine are past the shared scan's eight and worth a GEMM pass
    assert _check_batch(rlr, oracle, ix, rows, qs[:2], 100).n_batches == 0
    assert _check_batch(rlr, oracle, ix, rows, qs[:4], 100).n_batches == 1
    assert _check_batch(rlr, oracle, ix, rows, qs[:9], 100).n_batches == 1
    ix.close()


def test_batched_mfma_fp16_rows_1024d_more_than_256_queries(rlr, oracle):
    n = 30_000
    rows = oracle.synth_rows(n, 1024, seed=113, f16=True)
    qs = np.stack([oracle.normalize(oracle.synth_query(1024, seed=900 + i)) for i in range(300)])
    ix = rlr.GpuIndex(1024, "f16")
    ix.fill_synthetic(n, seed=113)
    ix.profile_read(reset=True)
    r, c = ix.search_topk(qs, 20)
    prof = ix.profile_read()
    assert prof.n_batches == 1 and prof.n_batch_queries == 300 and prof.n_batch_fallbacks == 0
    for i in range(0, 300, 7):  # spot-check every 7th query against the oracle (CPU time)
        wr, wc = oracle_topk(oracle, rows, qs[i], 20)
        assert np.array_equal(r[i], wr) and np.array_equal(bits(c[i]), bits(wc)), i
    # and the batched path agrees with the single-query path on every query
    for i in range(0, 300, 31):
        r1, c1 = ix.search_topk(qs[i], 20)
        assert np.array_equal(r1[0], r[i]) and np.array_equal(bits(c1[0]), bits(c[i]))
    ix.close()


def test_batched_mfma_falls_back_on_duplicate_flood(rlr, oracle):
    base = oracle.synth_rows(5000, 768, seed=114)
    q0 = oracle.synth_query(768, seed=115)
    dup = oracle.normalize(q0 + np.float32(0.01) * base[0])
    rows = np.concatenate([base, np.repeat(dup[None, :], 9000, axis=0)])
    qs = np.stack([oracle.normalize(q0)] + [oracle.normalize(oracle.synth_query(768, seed=950 + i)) for i in range(19)])
    ix = make_index(rlr, rows)
    prof = _check_batch(rlr, oracle, ix, rows, qs, 10)
    assert prof.n_batch_fallbacks >= 1          # query 0 overflows its band and is re-run alone
    ix.close()


@pytest.mark.parametrize("dim", [768, 1024, 128, 1600])
def test_single_query_nomination_over_the_image_is_exact(rlr, oracle, dim):
    """enable_batch_image(single_query=True): the scan streams the binary16 image, the wider band is re-scored from
    the f32 rows -- rows and scores identical to the oracle, for ragged row counts and after mutations"""
    for n, k in ((4099, 25), (300, 300), (20000, 100)):
        rows = oracle.synth_rows(n, dim, seed=900 + dim + n, n_clusters=11)
        ix = make_index(rlr, rows)
        ix.enable_batch_image(True, single_query=True)
        for qi in range(2):
            qn = oracle.normalize(oracle.synth_query(dim, seed=950 + dim + qi))
            r, c = ix.search_topk(qn, k)
            wr, wc = oracle_topk(oracle, rows, qn, k)
            assert np.array_equal(r[0], wr), (dim, n)
            assert np.array_equal(bits(c[0]), bits(wc)), (dim, n)
        if n == 4099:
            extra = oracle.synth_rows(300, dim, seed=77 + dim)
            ix.append(extra)
            ix.delete_rows([0, 255, 256, 4100])
            cur = np.delete(np.concatenate([rows, extra]), [0, 255, 256, 4100], axis=0)
            qn = oracle.normalize(extra[7] + oracle.synth_query(dim, seed=5) * np.float32(0.2))
            r, c = ix.search_topk(qn, k)
            wr, wc = oracle_topk(oracle, cur, qn, k)
            assert np.array_equal(r[0], wr) and np.array_equal(bits(c[0]), bits(wc))
            # a few queries at once take the per-query pipelines (below the batch threshold) -- same scan
            qs = np.stack([oracle.normalize(oracle.synth_query(dim, seed=960 + i)) for i in range(3)])
            r3, c3 = ix.search_topk(qs, k)
            for i in range(3):
                wr, wc = oracle_topk(oracle, cur, qs[i], k)
                assert np.array_equal(r3[i], wr) and np.array_equal(bits(c3[i]), bits(wc))
        ix.close()


@pytest.mark.parametrize("dim", [768, 1024, 64, 400, 384, 1536, 512, 1280])
def test_single_query_nomination_over_the_8bit_copy_is_exact(rlr, oracle, dim):
    """enable_batch_image(q8=True): one byte per element + a per-row scale; the band comes from the stored row error
    norms (Cauchy-Schwarz), the re-score from the f32 rows -- rows and scores identical to the oracle, including
    unnormalised queries, NaN rows, mutations; an Inf row switches the index back to the f32 scan"""
    for n, k in ((4099, 25), (300, 300), (20000, 100)):
        rows = oracle.synth_rows(n, dim, seed=1900 + dim + n, n_clusters=11)
        if n == 4099:
            rows[17, 3] = np.nan
            rows[18] = 0
        ix = make_index(rlr, rows)
        ix.enable_batch_image(False, q8=True)
        for qi in range(3):
            qn = oracle.normalize(oracle.synth_query(dim, seed=1950 + dim + qi))
            if qi == 2:
                qn = (qn * np.float32(3.5)).astype(np.float32)       # not unit norm: the band scales with ||q||
            r, c = ix.search_topk(qn, k)
            wr, wc = oracle_topk(oracle, rows, qn, k)
            assert np.array_equal(r[0], wr), (dim, n, qi)
            assert np.array_equal(bits(c[0]), bits(wc)), (dim, n, qi)
        if n == 4099:
            extra = oracle.synth_rows(300, dim, seed=177 + dim) * np.float32(2.0)   # longer rows: larger scales / errors
            ix.append(extra)
            ix.delete_rows([0, 255, 256, 4100])
            cur = np.delete(np.concatenate([rows, extra]), [0, 255, 256, 4100], axis=0)
            qs = np.stack([oracle.normalize(oracle.synth_query(dim, seed=1960 + i)) for i in range(3)])
            r3, c3 = ix.search_topk(qs, k)
            for i in range(3):
                wr, wc = oracle_topk(oracle, cur, qs[i], k)
                assert np.array_equal(r3[i], wr) and np.array_equal(bits(c3[i]), bits(wc)), (dim, i)
            cur2 = cur.copy()
            cur2[5, 0] = np.inf                                     # an Inf row: the index falls back to the f32 scan
            ix.upload(cur2)
            r, c = ix.search_topk(qs[0], k)
            wr, wc = oracle_topk(oracle, cur2, qs[0], k)
            assert np.array_equal(r[0], wr) and np.array_equal(bits(c[0]), bits(wc))
        ix.close()


@pytest.mark.parametrize("dim,nq", [(768, 2), (768, 8), (1024, 3), (256, 5), (512, 7)])
def test_small_batches_share_one_scan(rlr, oracle, monkeypatch, dim, nq):
    """2..8 queries over f32 rows: scan_multi_kernel reads the rows once for all of them, the batched select and
    finish follow; rows with duplicates / zeros / NaN included"""
    monkeypatch.setenv("RLR_BATCH_MIN", "2")          # take the batched pipeline whatever the cost model says
    n, k = 30_011, 40
    rows = oracle.synth_rows(n, dim, seed=3300 + dim + nq, n_clusters=9)
    rows[5] = rows[3]
    rows[n - 1] = rows[3]
    rows[7] = 0
    rows[9, 0] = np.nan
    ix = make_index(rlr, rows)
    qs = np.stack([oracle.normalize(oracle.synth_query(dim, seed=3350 + i)) for i in range(nq)])
    qs[0] = oracle.normalize(rows[3].copy())
    ix.profile_read(reset=True)
    r, c = ix.search_topk(qs, k)
    prof = ix.profile_read()
    assert prof.n_batches == 1 and prof.n_batch_fallbacks == 0
    for i in range(nq):
        wr, wc = oracle_topk(oracle, rows, qs[i], k)
        assert np.array_equal(r[i], wr), (dim, nq, i)
        assert np.array_equal(bits(c[i]), bits(wc)), (dim, nq, i)
    ix.close()


@pytest.mark.parametrize("dim", [1024, 768, 1536, 2048])
def test_8bit_copy_over_binary16_rows(rlr, oracle, dim):
    """the 8-bit nomination copy also serves f16-typed indexes (half of their scan bytes)"""
    n, k = 9001, 60
    rows = oracle.synth_rows(n, dim, seed=4100 + dim, n_clusters=7, f16=True)
    ix = make_index(rlr, rows, dtype="f16")
    ix.enable_batch_image(False, q8=True)
    for qi in range(3):
        qn = oracle.normalize(oracle.synth_query(dim, seed=4150 + qi))
        r, c = ix.search_topk(qn, k)
        wr, wc = oracle_topk(oracle, rows, qn, k)
        assert np.array_equal(r[0], wr) and np.array_equal(bits(c[0]), bits(wc)), (dim, qi)
    ix.close()


def test_8bit_scan_kernels_agree(rlr, oracle):
    """the lane-packed 8-bit scan (default where dim/16 times 2, 4 or 8 is a multiple of 64) and the one-row-per-load kernel (RLR_Q8_PACKED=0, read
    once per process, so in a child) both end in the oracle's answer on a corpus with ragged group tails"""
    import os
    import subprocess
    import sys
    code = (
        "import importlib, numpy as np, sys; sys.path.insert(0, '.')\n"
        "rlr = importlib.import_module('rust-local-rag_amd'); from oracle import oracle as O\n"
        "for dim in (768, 384, 1536, 512, 256, 128, 640, 1280):\n"
        "    for n in (1, 7, 33, 4099):\n"
        "        rows = O.synth_rows(n, dim, seed=77 + n + dim, n_clusters=3)\n"
        "        ix = rlr.GpuIndex(dim); ix.upload(rows); ix.enable_batch_image(False, q8=True)\n"
        "        q = O.normalize(O.synth_query(dim, seed=5 + n))\n"
        "        r, c = ix.search_topk(q, 10)\n"
        "        s = O.scan(rows, q); o = np.argsort(-s, kind='stable')[:10]\n"
        "        assert np.array_equal(r[0], o.astype(np.uint64)) and np.array_equal(c[0].view(np.uint32), s[o].view(np.uint32)), (dim, n)\n"
        "        ix.close()\n"
        "print('agree')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for packed in ("1", "0"):
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, RLR_Q8_PACKED=packed),
                             capture_output=True, text=True, timeout=240)
        assert out.returncode == 0 and "agree" in out.stdout, (packed, out.stdout[-1000:], out.stderr[-3000:])


def test_batched_floor_from_a_sample_rank_hands_back_what_it_cannot_certify(rlr, oracle):
    """run_batched takes the floor for the corpus pass from a rank of the sample ABOVE the k-th (rank k/4 of a sample a
    quarter the size); a query with fewer than k rows above that rank's score -- too few candidates, or a guard band that
    would reach below the floor -- must come back through the single-query path with the oracle's answer.  The rank is
    kept six deviations clear of that case, so it is forced low here (RLR_BATCH_RANK_FORCE, read once per process: a
    child), with the default and with a wide guard band."""
    import os
    import subprocess
    import sys
    code = (
        "import importlib, numpy as np, sys; sys.path.insert(0, '.')\n"
        "rlr = importlib.import_module('rust-local-rag_amd'); from oracle import oracle as O\n"
        "n, dim, k = 200_000, 768, 100\n"
        "rows = O.synth_rows(n, dim, seed=112)\n"
        "ix = rlr.GpuIndex(dim); ix.fill_synthetic(n, seed=112)\n"
        "qs = np.stack([O.normalize(O.synth_query(dim, seed=800 + i)) for i in range(24)])\n"
        "back = []\n"
        "for eps in (-1.0, 0.02):\n"
        "    ix.profile_read(reset=True)\n"
        "    r, c = ix.search_topk(qs, k, guard_eps=eps)\n"
        "    p = ix.profile_read()\n"
        "    assert p.n_batches == 1, p\n"
        "    back.append(int(p.n_batch_fallbacks))\n"
        "    for i in range(len(qs)):\n"
        "        s = O.scan(rows, qs[i]); o = np.argsort(-s, kind='stable')[:k]\n"
        "        assert np.array_equal(r[i], o.astype(np.uint64)) and np.array_equal(c[i].view(np.uint32), s[o].view(np.uint32)), (eps, i)\n"
        "print('handed back', back)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = {}
    for rank in ("8", "30", "0"):
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, RLR_BATCH_RANK_FORCE=rank),
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "handed back" in out.stdout, (rank, out.stdout[-1000:], out.stderr[-3000:])
        seen[rank] = eval(out.stdout.strip().splitlines()[-1].split("handed back", 1)[1])
    assert seen["8"][0] == 24 and seen["8"][1] == 24, seen     # 8 x 3 rows above the floor's rank: every query goes back
    assert 0 < seen["30"][0] <= 24, seen                        # about the mean of the sample's share of the best 99
    assert seen["0"] == [0, 0], seen                            # the rank the library picks by itself


def test_merge_topk_kernel_matches_torch_merge_and_global_oracle(rlr, oracle):
    """Four shards searched one after the other on the one GPU, their packed results laid out as an
    all-gather would deliver them, merged by rlr_merge_topk: must equal the global oracle and the
    torch merge used by the gloo test."""
    import ctypes as C
    import importlib
    import torch
    sharded = importlib.import_module("rust-local-rag_amd.sharded")
    n_total, dim, k, world, nq = 9001, 768, 25, 4, 3
    rows = oracle.synth_rows(n_total, dim, seed=321)
    lo1 = sharded.shard_range(n_total, 1, world)[0]
    rows[lo1 + 7] = rows[11]  # exact cross-shard tie
    qs = np.stack([oracle.normalize(rows[11])] + [oracle.normalize(oracle.synth_query(dim, seed=330 + i)) for i in range(nq - 1)])
    gathered = torch.zeros((world, nq, k), dtype=torch.int64, device="cuda")
    bases = []
    for r in range(world):
        lo, hi = sharded.shard_range(n_total, r, world)
        bases.append(lo)
        ix = make_index(rlr, rows[lo:hi])
        ix.search_topk_device(qs, k, gathered[r].data_ptr())
        ix.close()
    torch.cuda.synchronize()
    bases_h = np.array(bases, dtype=np.uint64)
    out_rows = np.zeros((nq, k), np.uint64)
    out_cos = np.zeros((nq, k), np.float32)
    out_n = np.zeros(nq, np.uint32)
    L = rlr.lib()
    st = L.rlr_merge_topk(0, C.c_void_p(gathered.data_ptr()), world, nq, k, bases_h.ctypes.data_as(C.POINTER(C.c_uint64)),
                          out_rows.ctypes.data_as(C.POINTER(C.c_uint64)), out_cos.ctypes.data_as(C.POINTER(C.c_float)),
                          out_n.ctypes.data_as(C.POINTER(C.c_uint32)), None)
    assert st == 0 and (out_n == k).all()
    t_rows, t_key = sharded.merge_packed(gathered.cpu(), torch.tensor(bases, dtype=torch.int64), k)
    assert np.array_equal(out_rows.astype(np.int64), t_rows.numpy())
    assert np.array_equal(bits(out_cos), bits(sharded.key_to_score(t_key.numpy())))
    for i in range(nq):
        wr, wc = oracle_topk(oracle, rows, qs[i], k)
        assert np.array_equal(out_rows[i], wr) and np.array_equal(bits(out_cos[i]), bits(wc))
    assert list(out_rows[0][:2]) == [11, lo1 + 7]  # the tie: lower global row first


# ---------------------------------------------------------------- batched MMR / batched engine
def test_mmr_select_batch_equals_single_calls(rlr, oracle):
    rows = oracle.synth_rows(6000, 768, seed=411, n_clusters=20)
    ix = make_index(rlr, rows)
    nq, P, k, lam = 70, 300, 100, 0.3   # 70 queries -> two passes of the 64-query chunking
    qs = np.stack([oracle.normalize(oracle.synth_query(768, seed=1000 + i)) for i in range(nq)])
    pr, pc = ix.search_topk(qs, P)
    sc = (np.float32(0.7) * pc).astype(np.float32)
    sizes = np.full(nq, P, np.uint32)
    sizes[3] = 17           # ragged pools
    sizes[5] = 1
    sizes[9] = 0
    order, mmr, n = ix.mmr_select_batch(pr, sc, sizes, k, lam)
    for q in range(nq):
        o1, m1 = ix.mmr_select(pr[q][:sizes[q]], sc[q][:sizes[q]], k, lam)
        assert n[q] == len(o1), q
        assert np.array_equal(order[q][:n[q]], o1), q
        assert np.array_equal(bits(mmr[q][1:n[q]]), bits(m1[1:]))
    # and one of them against the oracle
    wo, _ = oracle.mmr(rows[pr[0].astype(np.int64)], sc[0], k, lam)
    assert np.array_equal(order[0][:n[0]], wo)
    ix.close()


@pytest.mark.parametrize("nq,k,lam", [(20, 10, 0.5), (18, 100, 0.7), (5, 5, 0.0)])
def test_engine_search_with_diversity_batch_matches_oracle(rlr, oracle, nq, k, lam):
    n, dim = 12000, 768
    eng = rlr.RagEngine(dim)
    eng.index.fill_synthetic(n, seed=421, n_clusters=30)
    eng._chunks = [rlr.DocumentChunk(str(i), "synthetic", "", i) for i in range(n)]
    rows = oracle.synth_rows(n, dim, seed=421, n_clusters=30)
    qs = np.stack([oracle.synth_query(dim, seed=1100 + i) for i in range(nq)])
    got = eng.search_with_diversity_batch(qs, k, lam)
    assert len(got) == nq
    for q in range(nq):
        wr, wc, we, _ = oracle.search_with_diversity(rows, qs[q], k, lam)
        assert [g.row for g in got[q]] == list(wr), q
        assert np.array_equal(bits([g.score for g in got[q]]), bits(wc))
        assert np.array_equal(bits([g.embedding_score for g in got[q]]), bits(we))
    eng.close()


def test_search_stage1_then_reranker_blend(rlr, oracle):
    """stage-1 candidates from the GPU + the host-side blend of rag_engine.rs:599-700 (row f2)."""
    rows = oracle.synth_rows(4000, 768, seed=501)
    eng, ids = build_engine(rlr, rows)
    stored = eng.index.fetch_rows(np.arange(4000))
    q = oracle.synth_query(768, seed=502)
    cands = eng.search(q, 5, stage=1)                        # 15 candidates for the reranker
    wr, wc, _, _ = oracle.search(stored, q, 5, stage=1)
    assert [c.row for c in cands] == list(wr)
    rng = np.random.default_rng(1)
    pick = rng.permutation(15)[:9]
    rel = rng.random(9).astype(np.float32)
    reranked = [(cands[i].chunk_id, float(rel[j])) for j, i in enumerate(pick)]
    got = eng.finish_with_reranker(cands, reranked, 5)
    oc, os_, orr, oh = oracle.blend(wr, wc, [wr[i] for i in pick], rel, 5)
    assert [g.row for g in got] == [int(wr[i]) for i in oc]
    assert np.array_equal(bits([g.score for g in got]), bits(os_))
    assert [g.reranker_score is not None for g in got] == list(oh)
    eng.close()


# ---------------------------------------------------------------- nomination image (batched path)
@pytest.mark.parametrize("dim,dtype", [(768, "f32"), (1024, "f16")])
def test_batch_image_same_results_and_tracks_mutation(rlr, oracle, dim, dtype):
    n = 9000
    rows = oracle.synth_rows(n, dim, seed=611, f16=(dtype == "f16"))
    qs = np.stack([oracle.normalize(oracle.synth_query(dim, seed=1300 + i)) for i in range(24)])
    ix = rlr.GpuIndex(dim, dtype)
    ix.upload(oracle.synth_rows(n, dim, seed=611))
    ix.enable_batch_image(True)
    prof = _check_batch(rlr, oracle, ix, rows, qs, 50)
    assert prof.n_batches == 1 and prof.n_batch_fallbacks == 0
    # append + delete: the image must follow the row-major master copy
    extra = oracle.synth_rows(700, dim, seed=612, f16=(dtype == "f16"))
    ix.append(oracle.synth_rows(700, dim, seed=612))
    dead = np.array([3, 255, 256, 4000, 9100], dtype=np.uint64)
    ix.delete_rows(dead)
    all_rows = np.concatenate([rows, extra])
    keep = np.setdiff1d(np.arange(n + 700), dead)
    prof = _check_batch(rlr, oracle, ix, all_rows[keep], qs, 50)
    assert prof.n_batches == 1 and prof.n_batch_fallbacks == 0
    # dropping the image falls back to the row-major GEMM with identical results
    ix.enable_batch_image(False)
    _check_batch(rlr, oracle, ix, all_rows[keep], qs[:16], 50)
    ix.close()


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_unnormalised_rows_and_queries_keep_the_guarantee(rlr, oracle, dtype):
    """rows of norm up to 40 stored as given and a query of norm 7 through the raw ABI: the guard bands are bounds for
    unit-norm operands, so the library has to widen them by |row|_max * |query| -- single queries, a batch through the
    matrix cores, the nomination image and the 8-bit copy must all still return the reference's rows and scores."""
    n, dim = 60_000, 768
    rng = np.random.default_rng(404)
    base = oracle.synth_rows(n, dim, seed=4040, n_clusters=30, f16=False)
    scale = rng.uniform(0.2, 40.0, size=(n, 1)).astype(np.float32)
    rows = (base * scale).astype(np.float32)
    if dtype == "f16":
        rows = oracle.round_f16(rows)
    qs = np.stack([oracle.synth_query(dim, seed=4100 + i) for i in range(24)])
    qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True) * 7.0).astype(np.float32)
    ix = rlr.GpuIndex(dim, dtype)
    ix.upload(rows[: n // 2])
    ix.append(rows[n // 2:])                                       # the recorded norm follows append as well
    for mode in ("plain", "image", "image_scan", "q8"):
        if mode == "q8":
            ix.enable_batch_image(False, q8=True)
        elif mode != "plain":
            ix.enable_batch_image(True, single_query=(mode == "image_scan"))
        r1, c1 = ix.search_topk(qs[0], 100)
        wr, wc = oracle_topk(oracle, rows, qs[0], 100)
        assert np.array_equal(r1[0], wr) and np.array_equal(bits(c1[0]), bits(wc)), mode
        rb, cb = ix.search_topk(qs, 50)
        for i in range(0, 24, 5):
            wr, wc = oracle_topk(oracle, rows, qs[i], 50)
            assert np.array_equal(rb[i], wr) and np.array_equal(bits(cb[i]), bits(wc)), (mode, i)
    ix.close()


def test_multi_index_rccl_exchange_world1_and_persistent_workers(rlr, oracle):
    """rlr_multi_set_exchange(m, 1): the partial top-k lists stay on the device, ncclAllGather (single-process
    communicator, here of one rank: the one-GPU box) + merge_topk_kernel -- same results as the host merge and as the
    oracle; duplicate devices are refused.  Then the persistent shard workers under concurrent callers."""
    import threading

    n, dim = 30_011, 768
    rows = oracle.synth_rows(n, dim, seed=7771, n_clusters=11)
    qs = np.stack([oracle.normalize(oracle.synth_query(dim, seed=7800 + i)) for i in range(6)])
    one = rlr.MultiGpuIndex(dim, [0])
    one.upload(rows)
    host = one.search_topk(qs, 100)
    one.set_exchange("rccl")
    got = one.search_topk(qs, 100)
    assert np.array_equal(got[0], host[0]) and np.array_equal(bits(got[1]), bits(host[1]))
    for i in range(len(qs)):
        wr, wc = oracle_topk(oracle, rows, qs[i], 100)
        assert np.array_equal(got[0][i], wr) and np.array_equal(bits(got[1][i]), bits(wc)), i
    r1, c1 = one.search_topk(qs[0], 7)                      # other shapes reuse / regrow the exchange buffers
    assert np.array_equal(r1[0], host[0][0][:7])
    one.set_exchange("host")
    one.close()
    two = rlr.MultiGpuIndex(dim, [0, 0])
    with pytest.raises(rlr.RlrError):
        two.set_exchange("rccl")                            # RCCL wants one rank per device
    two.close()
    # four shards on one GPU, eight host threads searching at once: every answer still the oracle's
    mi = rlr.MultiGpuIndex(dim, [0, 0, 0, 0])
    mi.upload(rows)
    want = [oracle_topk(oracle, rows, q, 20) for q in qs]
    bad = []

    def hammer(t):
        for rep in range(12):
            i = (t + rep) % len(qs)
            r, c = mi.search_topk(qs[i], 20)
            if not (np.array_equal(r[0], want[i][0]) and np.array_equal(bits(c[0]), bits(want[i][1]))):
                bad.append((t, rep))

    th = [threading.Thread(target=hammer, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not bad, bad
    mi.close()


@pytest.mark.parametrize("dim,dtype,n,nq,k", [
    (768, "f32", 70_001, 130, 100),    # one query block, partly filled; ragged last row tile
    (768, "f32", 300_000, 256, 100),   # sample + bootstrap + filtered main pass, every workgroup several units
    (1024, "f16", 45_300, 300, 20),    # two query blocks (the second ragged): units of one tile on neighbouring workgroups
    (384, "f32", 33_000, 520, 10),     # three query blocks; dim / 64 = 6 (not a multiple of 4: 8-phase kernel only)
    (768, "f32", 5_000, 140, 100),     # fewer row tiles than workgroups
])
def test_batch_image_8phase_kernel(rlr, oracle, dim, dtype, n, nq, k):
    """More than 128 queries over the nomination image take the persistent 8-phase LDS-DMA GEMM (gemm8_kernel):
    rows and score bits equal to the oracle on a sample of the queries and to the single-query path on all."""
    f16 = dtype == "f16"
    rows = oracle.synth_rows(n, dim, seed=640, f16=f16)
    qs = np.stack([oracle.normalize(oracle.synth_query(dim, seed=1500 + i)) for i in range(nq)])
    ix = rlr.GpuIndex(dim, dtype)
    ix.fill_synthetic(n, seed=640)
    assert np.array_equal(ix.fetch_rows(np.arange(0, n, 997)).view(np.uint32), rows[::997].view(np.uint32))
    ix.enable_batch_image(True)
    ix.profile_read(reset=True)
    r, c = ix.search_topk(qs, k)
    prof = ix.profile_read()
    assert prof.n_batches == 1 and prof.n_batch_queries == nq and prof.n_batch_fallbacks == 0
    for i in list(range(0, nq, max(1, nq // 12))) + [nq - 1]:
        wr, wc = oracle_topk(oracle, rows, qs[i], k)
        assert np.array_equal(r[i], wr), f"query {i}: rows differ"
        assert np.array_equal(bits(c[i]), bits(wc)), f"query {i}: scores differ"
    ix.enable_batch_image(False)
    for i in range(nq):
        r1, c1 = ix.search_topk(qs[i], k)
        assert np.array_equal(r1[0], r[i]) and np.array_equal(bits(c1[0]), bits(c[i])), i
    ix.close()


def test_batch_image_8phase_kernel_survives_a_candidate_flood(rlr, oracle):
    """9000 copies of one chunk right at the top of one query's scores: its candidate list (and the workgroup-local
    LDS list of the tiles holding the copies) overflow; that query goes back to the single-query pipeline, the other
    queries of the batch are unaffected -- results still identical to the oracle."""
    base = oracle.synth_rows(30_000, 768, seed=641)
    q0 = oracle.synth_query(768, seed=642)
    dup = oracle.normalize(q0 + np.float32(0.01) * base[0])
    rows = np.concatenate([base[:20_000], np.repeat(dup[None, :], 9000, axis=0), base[20_000:]])
    qs = np.stack([oracle.normalize(q0)] + [oracle.normalize(oracle.synth_query(768, seed=1700 + i)) for i in range(149)])
    ix = make_index(rlr, rows)
    ix.enable_batch_image(True)
    ix.profile_read(reset=True)
    r, c = ix.search_topk(qs, 10)
    prof = ix.profile_read()
    assert prof.n_batches == 1 and prof.n_batch_fallbacks >= 1
    for i in [0, 1, 2, 77, 149]:
        wr, wc = oracle_topk(oracle, rows, qs[i], 10)
        assert np.array_equal(r[i], wr) and np.array_equal(bits(c[i]), bits(wc)), i
    ix.close()


# ---------------------------------------------------------------- one process, several shards
def test_multi_index_three_shards_on_one_gpu(rlr, oracle):
    """rlr_multi_* with three shards that all live on GPU 0: concurrent per-shard searches from
    three host threads, host merge, cross-shard ties, score/fetch routing, MMR over a pool that
    spans shards."""
    n, dim = 10_001, 768
    rows = oracle.synth_rows(n, dim, seed=711, n_clusters=15)
    rows[7000] = rows[12]                     # exact tie across shards 0 and 2
    mi = rlr.MultiGpuIndex(dim, [0, 0, 0])
    mi.upload(rows)
    assert len(mi) == n
    qs = np.stack([oracle.normalize(rows[12])] + [oracle.normalize(oracle.synth_query(dim, seed=1400 + i)) for i in range(4)])
    r, c = mi.search_topk(qs, 40)
    for i in range(len(qs)):
        wr, wc = oracle_topk(oracle, rows, qs[i], 40)
        assert np.array_equal(r[i], wr) and np.array_equal(bits(c[i]), bits(wc)), i
    assert list(r[0][:2]) == [12, 7000]
    pick = np.array([0, 3333, 3334, 6667, 6668, 10000, 5], dtype=np.uint64)   # shard boundaries
    assert np.array_equal(bits(mi.fetch_rows(pick)), bits(rows[pick.astype(np.int64)]))
    want = np.array([oracle.dot(qs[1], rows[int(p)]) for p in pick], np.float32)
    assert np.array_equal(bits(mi.score_rows(qs[1], pick)), bits(want))
    pr, pc = mi.search_topk(qs[1], 300)
    sc = (np.float32(0.7) * pc[0]).astype(np.float32)
    order, _ = mi.mmr_select(pr[0], sc, 100, 0.3)
    worder, _ = oracle.mmr(rows[pr[0].astype(np.int64)], sc, 100, 0.3)
    assert np.array_equal(order, worder)
    # batched queries go through the MFMA path on every shard
    qb = np.stack([oracle.normalize(oracle.synth_query(dim, seed=1500 + i)) for i in range(20)])
    rb, cb = mi.search_topk(qb, 10)
    for i in (0, 7, 19):
        wr, wc = oracle_topk(oracle, rows, qb[i], 10)
        assert np.array_equal(rb[i], wr) and np.array_equal(bits(cb[i]), bits(wc))
    mi.close()


def test_multi_index_reports_a_shard_failure_with_its_message(rlr):
    """a failure inside a shard's worker thread reaches the caller with the worker's message (rlr_last_error is
    per thread): 2 x 100 M rows of 8 KiB cannot be allocated, and the index stays usable afterwards"""
    mi = rlr.MultiGpuIndex(2048, [0, 0])
    with pytest.raises(rlr.RlrError) as err:
        mi.fill_synthetic(200_000_000, seed=1)
    assert "shard 0:" in str(err.value) and "failed" in str(err.value), str(err.value)
    mi.fill_synthetic(2000, seed=1)
    r, c = mi.search_topk(np.ones(2048, np.float32) / np.float32(np.sqrt(2048.0)), 5)
    assert r.shape == (1, 5) and np.all(np.diff(c[0]) <= 0)
    mi.close()
    # same on the calling thread: the failed allocation must not linger as a stale HIP error for the next launch check
    one = rlr.GpuIndex(2048)
    with pytest.raises(rlr.RlrError) as err:
        one.fill_synthetic(100_000_000, seed=1)
    assert err.value.status == rlr._native.RLR_E_OOM
    one.fill_synthetic(2000, seed=1)
    r1, c1 = one.search_topk(np.ones(2048, np.float32) / np.float32(np.sqrt(2048.0)), 5)
    assert np.array_equal(r1[:, :5], r) and np.array_equal(bits(c1), bits(c))
    one.close()


# ---------------------------------------------------------------- rarely taken paths
def test_multi_query_band_overflow_rescans(rlr, oracle):
    """fewer than 16 queries (looped single-query pipeline) where one query's band overflows:
    the large-candidate path has to re-scan because the score buffer holds a later query."""
    base = oracle.synth_rows(3000, 768, seed=801)
    q0 = oracle.synth_query(768, seed=802)
    dup = oracle.normalize(q0 + np.float32(0.01) * base[0])
    rows = np.concatenate([base, np.repeat(dup[None, :], 5000, axis=0)])
    qs = np.stack([oracle.normalize(q0)] + [oracle.normalize(oracle.synth_query(768, seed=810 + i)) for i in range(4)])
    ix = make_index(rlr, rows)
    r, c = ix.search_topk(qs, 12)
    for i in range(5):
        wr, wc = oracle_topk(oracle, rows, qs[i], 12)
        assert np.array_equal(r[i], wr) and np.array_equal(bits(c[i]), bits(wc)), i
    assert ix.profile_read().n_retries >= 1
    ix.close()


def test_k_larger_than_the_lds_sort(rlr, oracle):
    rows = oracle.synth_rows(12000, 256, seed=803)
    qn = oracle.normalize(oracle.synth_query(256, seed=804))
    ix = make_index(rlr, rows)
    r, c = ix.search_topk(qn, 5000)          # > 4096: global-memory bitonic path
    wr, wc = oracle_topk(oracle, rows, qn, 5000)
    assert np.array_equal(r[0], wr) and np.array_equal(bits(c[0]), bits(wc))
    ix.close()


@pytest.mark.parametrize("dim", [200, 72, 1536])
def test_fp16_rows_generic_dims(rlr, oracle, dim):
    rows = oracle.synth_rows(2500, dim, seed=805 + dim, f16=True)
    qn = oracle.normalize(oracle.synth_query(dim, seed=806))
    ix = rlr.GpuIndex(dim, "f16")
    ix.reserve(5000)
    ix.upload(oracle.synth_rows(2500, dim, seed=805 + dim))
    r, c = ix.search_topk(qn, 33)
    wr, wc = oracle_topk(oracle, rows, qn, 33)
    assert np.array_equal(r[0], wr) and np.array_equal(bits(c[0]), bits(wc))
    assert np.array_equal(bits(ix.fetch_rows(wr[:5])), bits(rows[wr[:5].astype(np.int64)]))
    ix.close()
