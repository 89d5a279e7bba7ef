"""RagEngine::search / search_with_diversity over a corpus sharded across several GPUs in ONE process, behind the C ABI
(rlr_multi_engine_*, include/rlr_engine.h): bit-identical to the oracle's search over the whole corpus -- f32 and binary16
rows, lexical pairs, weight overrides, rounding-tie chains at the fetch boundary, batches with cross-shard MMR -- on one
GPU with repeated device ids (host-merge exchange, the winner rows still travel device to device) and at world = 1 with
the RCCL exchange.  Reference: rag_engine.rs:470-475, :717-759; SURVEY.md 8(e)."""
import importlib

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def check(hits, want, what):
    wr, wc, we, wl = want
    assert [int(r) for r in hits["row"]] == [int(r) for r in wr], what
    assert np.array_equal(bits(hits["score"]), bits(wc)), what
    assert np.array_equal(bits(hits["embedding_score"]), bits(we)), what
    assert np.array_equal(bits(hits["lexical_score"]), bits(wl)), what
    assert np.array_equal(bits(hits["initial_score"]), bits(wc)), what


def corpus(oracle, n, dim, f16, seed):
    rows = oracle.synth_rows(n, dim, seed=seed, n_clusters=11, f16=f16)
    rows[n - 1] = rows[3]            # exact ties across the first and the last shard
    rows[n // 2 + 5] = rows[3]
    return rows


@pytest.mark.parametrize("dtype,devices,exchange", [("f32", [0, 0, 0], "host"), ("f16", [0, 0, 0, 0, 0], "host"),
                                                    ("f32", [0], "rccl"), ("f16", [0], "rccl")])
def test_multi_engine_search_matches_the_oracle(rlr, oracle, dtype, devices, exchange):
    f16 = dtype == "f16"
    n, dim = 9001, (1024 if f16 else 768)
    rows = corpus(oracle, n, dim, f16, seed=3101)
    mi = rlr.MultiGpuIndex(dim, devices, dtype)
    mi.upload(rows)
    mi.set_exchange(exchange)
    rng = np.random.default_rng(5)
    qs = [rows[3] * np.float32(3.0)] + [oracle.synth_query(dim, seed=3200 + i) for i in range(3)]   # raw (unnormalised) queries
    lex_rows = np.concatenate([rng.choice(n, size=40, replace=False), [3, n - 1, n + 7]]).astype(np.uint64)  # one beyond the index
    lex_scores = (rng.random(lex_rows.size) * 9).astype(np.float32)
    lex = list(zip(lex_rows.tolist(), lex_scores.tolist()))
    for qi, q in enumerate(qs):
        for k in (1, 10, 100):
            check(mi.engine_search(q, k), oracle.search(rows, q, k), ("search", qi, k))
            check(mi.engine_search(q, k, stage=1), oracle.search(rows, q, k, stage=1), ("stage1", qi, k))
            check(mi.engine_search(q, k, lex_rows=lex_rows, lex_scores=lex_scores), oracle.search(rows, q, k, lex=lex),
                  ("hybrid", qi, k))
        for k, lam in ((5, 0.3), (100, 0.7), (7, 1.0), (10, 0.0), (0, 0.4)):
            check(mi.engine_search_with_diversity(q, k, lam), oracle.search_with_diversity(rows, q, k, lam), ("mmr", qi, k, lam))
            check(mi.engine_search_with_diversity(q, k, lam, lex_rows=lex_rows, lex_scores=lex_scores),
                  oracle.search_with_diversity(rows, q, k, lam, lex=lex), ("mmr+lex", qi, k, lam))
    # weight overrides: w_e = 0 (no scan: lowest rows win the all-zero tie), w_e tiny (distinct cosines round to one combined
    # score: the tie chain reaches the fetch boundary and the fetch is widened)
    for w_e in (0.0, 1e-40, 3e-39):
        w = rlr.QueryWeights(embedding=w_e)
        check(mi.engine_search(qs[1], 10, weights=w, lex_rows=lex_rows, lex_scores=lex_scores),
              oracle.search(rows, qs[1], 10, w_e=w_e, lex=lex), ("w_e", w_e))
        check(mi.engine_search_with_diversity(qs[2], 20, 0.5, weights=w), oracle.search_with_diversity(rows, qs[2], 20, 0.5, w_e=w_e),
              ("w_e mmr", w_e))
    r, s = mi.engine_embedding_candidates(qs[1], 200)
    wr, we = oracle.embedding_candidates(rows, qs[1], 200)
    assert np.array_equal(r, wr) and np.array_equal(bits(s), bits(we))
    st = mi.stats()
    if exchange == "rccl":
        assert st["n_topk_rccl"] > 0 and st["n_topk_rccl_fell_back"] == 0, st
    else:
        assert st["n_topk_host_merge"] > 0 and st["n_topk_rccl"] == 0, st
        assert st["n_mmr_exchanges"] > 0 and st["mmr_exchange_bytes"] > 0, st     # the winner rows moved device to device
    mi.close()


@pytest.mark.parametrize("dtype,devices", [("f32", [0, 0, 0]), ("f16", [0, 0, 0, 0])])
def test_multi_engine_batch_with_cross_shard_mmr(rlr, oracle, dtype, devices):
    """BASELINE config 5's shape in small: a batch of queries, top-k with MMR, pools spanning every shard; each query equal
    to the oracle's search_with_diversity over the whole corpus (and the plain batch to its search)."""
    f16 = dtype == "f16"
    n, dim, nq = 9_007, (1024 if f16 else 768), 21
    rows = corpus(oracle, n, dim, f16, seed=3301)
    mi = rlr.MultiGpuIndex(dim, devices, dtype)
    mi.upload(rows)
    qs = np.stack([oracle.synth_query(dim, seed=3400 + i) for i in range(nq)])
    qs[0] = rows[3]
    if dtype == "f16":
        mi.enable_batch_image(1)           # the shards' batched searches nominate over their binary16 images
    for k, lam in ((10, 0.7), (100, 0.3), (5, 0.0)):
        got = mi.engine_search_with_diversity_batch(qs, k, lam)
        assert len(got) == nq
        for i in range(nq):
            check(got[i], oracle.search_with_diversity(rows, qs[i], k, lam), ("batch", i, k, lam))
    # tiny embedding weight: rounding-tie chains reach the fetch boundary, those queries take the single-query path
    w = rlr.QueryWeights(embedding=1e-40)
    got = mi.engine_search_with_diversity_batch(qs[:20], 5, 0.5, weights=w)
    for i in range(20):
        check(got[i], oracle.search_with_diversity(rows, qs[i], 5, 0.5, w_e=1e-40), ("batch w_e", i))
    # the raw batched MMR entry: ragged pool sizes, an empty pool, pools living on one shard only
    rng = np.random.default_rng(9)
    P, m = 64, 9
    prow = rng.integers(0, n, size=(m, P)).astype(np.uint64)
    prow[2] = np.arange(P)                      # all on shard 0
    psc = np.sort(rng.random((m, P)).astype(np.float32), axis=1)[:, ::-1].copy()
    sizes = np.array([64, 1, 64, 0, 33, 64, 2, 17, 64], np.uint32)
    order, mmr, nsel = mi.mmr_select_batch(prow, psc, sizes, 20, 0.4)
    for q in range(m):
        sz = int(sizes[q])
        worder, wmmr = oracle.mmr(rows[prow[q, :sz].astype(np.int64)], psc[q, :sz], 20, 0.4) if sz else (np.zeros(0, np.uint32), np.zeros(0, np.float32))
        assert int(nsel[q]) == len(worder) and np.array_equal(order[q, :nsel[q]], worder), q
        assert np.array_equal(bits(mmr[q, 1:nsel[q]]), bits(wmmr[1:])), q
    mi.close()


def test_multi_engine_search_text_matches_the_oracle(rlr, oracle):
    """query text -> GPU BM25 over the global rows -> blend over the shards (rlr_multi_engine_search_text) == the oracle's
    search with the oracle's BM25 pairs (oracle/lexical.py)"""
    lexmod = importlib.import_module("rust-local-rag_amd.lexical")
    from oracle import lexical as OL

    rng = np.random.default_rng(31)
    n, dim = 5000, 768
    rows = oracle.synth_rows(n, dim, seed=3501, n_clusters=7)
    vocab = np.array([f"w{i:03d}x" for i in range(300)])
    zipf = 1.0 / np.arange(1, 301)
    zipf /= zipf.sum()
    texts = [" ".join(vocab[rng.choice(300, size=int(rng.integers(3, 25)), p=zipf)]) for _ in range(n)]
    g = lexmod.LexicalIndex(0)
    o = OL.LexicalIndex()
    for r, t in enumerate(texts):
        g.add_chunk(r, t)
        o.add_chunk(r, t, rank=r)
    mi = rlr.MultiGpuIndex(dim, [0, 0, 0])
    mi.upload(rows)
    for i in range(6):
        q = oracle.synth_query(dim, seed=3600 + i)
        text = " ".join(vocab[rng.choice(300, size=int(rng.integers(1, 5)), p=zipf)])
        tokens = " ".join(lexmod.tokenize(text))
        for k, lam in ((10, 0.0), (10, 0.3), (100, 0.7)):
            k_seen = k if lam == 0.0 else max(3 * k, k + 10)
            pairs = o.score(text, 5 * max(k_seen, 1), keep_zero=False)
            want = oracle.search_with_diversity(rows, q, k, lam, lex=pairs)
            check(mi.engine_search_text(g, q, tokens, k, lam), want, ("text", i, k, lam))
    g.close()
    mi.close()


def test_multi_engine_concurrent_callers(rlr, oracle):
    """the reference serves searches from several worker threads under a read lock (mcp_server.rs:89, :377): eight callers on
    a four-shard index, every answer the oracle's (more callers than exchange workspaces: some wait)"""
    import threading

    n, dim = 6000, 768
    rows = oracle.synth_rows(n, dim, seed=3701, n_clusters=5)
    mi = rlr.MultiGpuIndex(dim, [0, 0, 0, 0])
    mi.upload(rows)
    qs = [oracle.synth_query(dim, seed=3800 + i) for i in range(6)]
    want = [oracle.search_with_diversity(rows, q, 10, 0.5) for q in qs]
    bad = []

    def worker(t):
        try:
            for rep in range(10):
                i = (t + rep) % len(qs)
                h = mi.engine_search_with_diversity(qs[i], 10, 0.5)
                if [int(r) for r in h["row"]] != [int(r) for r in want[i][0]] or not np.array_equal(bits(h["score"]), bits(want[i][1])):
                    bad.append((t, rep))
        except Exception as e:  # noqa: BLE001
            bad.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not bad, bad[:4]
    mi.close()


@pytest.mark.parametrize("exchange", ["host", "rccl"])
def test_multi_engine_on_two_distinct_devices(rlr, oracle, exchange):
    """The cross-device half that one GPU cannot exercise: shards on devices 0 and 1 -- peer mapping at create time,
    hipMemcpyPeerAsync in the winner-row exchange, a two-rank ncclAllGather in the RCCL form.  Skipped on a one-GPU box (the
    driver's multi-GPU node runs it)."""
    if rlr.device_count() < 2:
        pytest.skip("needs two GPUs")
    n, dim = 9001, 768
    rows = corpus(oracle, n, dim, False, seed=3301)
    mi = rlr.MultiGpuIndex(dim, [0, 1], "f32")
    mi.upload(rows)
    mi.set_exchange(exchange)
    qs = [oracle.synth_query(dim, seed=3400 + i) for i in range(3)]
    for qi, q in enumerate(qs):
        check(mi.engine_search(q, 100), oracle.search(rows, q, 100), ("search", qi))
        check(mi.engine_search_with_diversity(q, 20, 0.7), oracle.search_with_diversity(rows, q, 20, 0.7), ("mmr", qi))
    st = mi.stats()
    assert st["n_mmr_exchanges"] > 0
    assert st["n_mmr_host_bounces"] == 0, st          # the peer copies worked (a bounce is correct too, but worth knowing)
    if exchange == "rccl":
        assert st["n_topk_rccl"] > 0, st
    mi.close()
