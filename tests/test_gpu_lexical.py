"""GPU BM25 (include/rlr_lexical.h) against the oracle restatement of LexicalIndex
(oracle/lexical.py): rows and scores bit-exact, both selection paths, mutations, and the hybrid
search end to end from query text."""
import importlib

import numpy as np
import pytest

from oracle import lexical as OL

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


VOCAB = [f"w{i:03d}x" for i in range(400)] + ["common", "frequent", "the", "of", "né", "Straße", "ÉTÉ"]


def make_texts(n, seed, lo=3, hi=40, common_every=0):
    rng = np.random.default_rng(seed)
    zipf = 1.0 / np.arange(1, len(VOCAB) + 1)
    zipf /= zipf.sum()
    texts = []
    for i in range(n):
        m = int(rng.integers(lo, hi))
        words = list(rng.choice(VOCAB, size=m, p=zipf))
        if common_every and i % common_every == 0:
            words.append("ubiquitous")
        texts.append(" ".join(words) + (", " if i % 3 else " - "))
    return texts


def build_pair(rlr, texts):
    lex = importlib.import_module("rust-local-rag_amd.lexical")
    g = lex.LexicalIndex(0)
    o = OL.LexicalIndex()
    for r, t in enumerate(texts):
        g.add_chunk(r, t)
        o.add_chunk(r, t, rank=r)
    return g, o


def check(g, o, query, limit):
    rows, sc = g.score(query, limit)
    want = o.score(query, limit if limit else 0, keep_zero=False)
    if limit == 0:
        want = want[:8192]
    assert [int(r) for r in rows] == [c for c, _ in want], (query, limit)
    assert np.array_equal(bits(sc), bits([s for _, s in want])), (query, limit)
    return len(rows)


def test_bm25_small_corpus_bit_exact(rlr):
    texts = make_texts(1500, seed=1)
    g, o = build_pair(rlr, texts)
    info = g.info()
    assert info["total_docs"] == o.total_docs and info["total_length"] == o.total_length
    assert info["n_terms"] == len(o.term_postings)
    assert info["n_postings"] == sum(len(p) for p in o.term_postings.values())
    n_hit = 0
    for q, lim in [("w000x", 25), ("w017x w101x", 50), ("W399X, w250x; w250x w003x", 500), ("unknown words only", 10),
                   ("né Straße été", 100), ("w005x", 0), ("the of", 1500), ("it is", 5), ("", 5),
                   ("w001x w002x w003x w004x w005x w006x w007x w008x w009x w010x w011x", 7)]:
        n_hit += check(g, o, q, lim)
    assert n_hit > 500
    g.close()


def test_bm25_select_path_many_postings_and_ties(rlr):
    """> 8192 touched documents -> radix select over the packed keys; short identical documents give
    massive exact score ties that must resolve to the lower row"""
    texts = make_texts(40000, seed=2, lo=2, hi=6, common_every=3)  # 13 334 documents hold "ubiquitous"
    texts += ["ubiquitous rare"] * 50 + ["rare"] * 3
    g, o = build_pair(rlr, texts)
    for q, lim in [("ubiquitous", 100), ("ubiquitous", 1500), ("ubiquitous rare w000x", 8192), ("w000x w001x common", 300),
                   ("ubiquitous", 0)]:
        n = check(g, o, q, lim)
        assert n == (lim if lim else 8192)  # every query here has more positive documents than that
    g.close()


def test_bm25_mutations_follow_row_compaction(rlr):
    lex = importlib.import_module("rust-local-rag_amd.lexical")
    texts = make_texts(300, seed=3)
    g, o = build_pair(rlr, texts)
    check(g, o, "w000x w001x", 40)
    # re-add replaces (:2107-2109); a chunk without tokens is not indexed
    g.add_chunk(5, "w000x w000x w000x brandnew")
    o.add_chunk(5, "w000x w000x w000x brandnew", rank=5)
    g.add_chunk(6, "a b")
    o.add_chunk(6, "a b", rank=6)
    assert g.contains(5) and not g.contains(6) and not g.contains(10_000)
    check(g, o, "w000x brandnew", 40)
    # delete rows: survivors move down, exactly like GpuIndex.delete_rows
    dead = [0, 7, 8, 150, 299]
    g.remove_rows(dead)
    o2 = OL.LexicalIndex()
    keep = [r for r in range(300) if r not in dead]
    cur = {r: texts[r] for r in range(300)}
    cur[5] = "w000x w000x w000x brandnew"
    cur[6] = "a b"
    for new_r, r in enumerate(keep):
        o2.add_chunk(new_r, cur[r], rank=new_r)
    assert g.info()["total_docs"] == o2.total_docs and g.info()["total_length"] == o2.total_length
    for q in ("w000x brandnew", "w002x w003x", "common"):
        check(g, o2, q, 60)
    g.clear()
    assert g.info()["total_docs"] == 0 and g.score("w000x", 5)[0].size == 0
    g.close()
    assert lex.tokenize("x") == []


def test_engine_hybrid_search_from_query_text(rlr, oracle):
    """RagEngine.search(query_text=...): GPU BM25 feeds the hybrid blend; same results as the oracle's
    search given the oracle's lexical pairs (rag_engine.rs:505-545)."""
    n, dim, k = 1200, 256, 10
    texts = make_texts(n, seed=4)
    rows = oracle.synth_rows(n, dim, seed=44)
    eng = rlr.RagEngine(dim)
    eng.add_document("doc.pdf", texts, rows)
    stored = eng.index.fetch_rows(np.arange(n))
    o = OL.LexicalIndex()
    for r, t in enumerate(texts):
        o.add_chunk(r, t, rank=r)
    q = oracle.synth_query(dim, seed=45)
    for text, div in (("w010x w020x common", 0.0), ("w001x", 0.0), ("w003x w004x", 0.4)):
        k_eff = k if div == 0.0 else max(3 * k, k + 10)
        pairs = [(c, float(s)) for c, s in o.score(text, 5 * k_eff, keep_zero=False)]
        got = eng.search_with_diversity(q, k, div, query_text=text)
        wr, wc, we, wl = oracle.search_with_diversity(stored, q, k, div, lex=pairs)
        assert [g_.row for g_ in got] == list(wr), text
        assert np.array_equal(bits([g_.score for g_ in got]), bits(wc)), text
        assert np.array_equal(bits([g_.lexical_score for g_ in got]), bits(wl)), text
    # removing a document drops its postings
    eng.add_document("other.pdf", ["solitaryterm appears here"], oracle.synth_rows(1, dim, seed=46))
    assert eng.lexical.score("solitaryterm", 5)[0].tolist() == [n]
    eng.remove_document("doc.pdf")
    # one document left: it moved to row 0 (idf of a term held by the only document is floored at 0)
    assert eng.lexical.contains(0) and not eng.lexical.contains(1) and eng.lexical.info()["total_docs"] == 1
    assert eng.lexical.score("solitaryterm", 5)[0].size == 0
    eng.close()


def test_gpu_bm25_against_the_hand_derived_vectors(rlr):
    """the GPU BM25 index against tests/golden/bm25_vectors.json (exact-arithmetic derivation, no oracle involved)"""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "bm25_vectors.json")) as f:
        doc = json.load(f)
    lex = importlib.import_module("rust-local-rag_amd.lexical")
    for case in doc["cases"]:
        g = lex.LexicalIndex(0)
        for r, tokens in enumerate(case["docs"]):
            g.add_tokens(r, tokens)
        for q in case["queries"]:
            rows, sc = g.score_tokens(q["terms"], q["limit"])
            assert [int(r) for r in rows] == q["rows"], (case["name"], q["terms"])
            assert [int(x) for x in bits(sc)] == q["score_bits"], (case["name"], q["terms"])
        g.close()


def test_bm25_concurrent_score_calls_use_their_own_workspaces(rlr):
    """Scoring calls from several host threads run side by side (one workspace and stream each, more callers than
    workspaces wait): every answer is the serial one, bit for bit, on both selection paths, and a mutation between
    two waves of calls is seen by all of them."""
    import threading

    texts = make_texts(30000, seed=11, lo=2, hi=10, common_every=2)  # "ubiquitous": 15 000 postings -> select path
    g, o = build_pair(rlr, texts)
    queries = [("ubiquitous", 100), ("w000x w001x", 50), ("w017x", 0), ("ubiquitous w003x common", 500),
               ("frequent the of", 25), ("w399x w398x w397x", 10)]
    want = {}
    for q, lim in queries:
        rows, sc = g.score(q, lim)
        check(g, o, q, lim)
        want[(q, lim)] = (rows.copy(), sc.copy())

    def wave(expect, n_threads=12, rounds=25):
        errors = []

        def worker(tid):
            try:
                for i in range(rounds):
                    q, lim = queries[(tid + i) % len(queries)]
                    rows, sc = g.score(q, lim)
                    er, es = expect[(q, lim)]
                    if not (np.array_equal(rows, er) and np.array_equal(bits(sc), bits(es))):
                        errors.append((tid, i, q, lim))
            except Exception as e:  # noqa: BLE001 -- reported below
                errors.append((tid, repr(e)))

        ts = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors[:5]

    wave(want)
    # grow the index past every workspace's accumulator and change the statistics
    extra = make_texts(12000, seed=12, lo=2, hi=10, common_every=2)
    for i, t in enumerate(extra):
        g.add_chunk(len(texts) + i, t)
        o.add_chunk(len(texts) + i, t, rank=len(texts) + i)
    want2 = {}
    for q, lim in queries:
        check(g, o, q, lim)
        rows, sc = g.score(q, lim)
        want2[(q, lim)] = (rows.copy(), sc.copy())
    assert any(not np.array_equal(want[k][0], want2[k][0]) for k in want)
    wave(want2)
    g.close()


def test_bm25_appends_rebuild_only_the_appended_segment(rlr):
    """An ingest loop that searches after every document: appended rows land in the second posting segment (no full
    rebuild), answers stay bit-equal to the oracle while statistics (df, average length) move under both segments;
    replacing an old row and an appended segment that outgrows main / 8 each force one full rebuild; removing rows does
    not (both segments are compacted and renumbered on the device)."""
    texts = make_texts(6000, seed=21, lo=30, hi=60)             # ~200 k postings in the main segment
    g, o = build_pair(rlr, texts)
    queries = [("w000x w001x", 50), ("common frequent w010x", 200), ("w017x", 0), ("né Straße w399x", 20)]
    for q, lim in queries:
        check(g, o, q, lim)
    seg = g.segments()
    assert seg["full_rebuilds"] == 1 and seg["append_rebuilds"] == 0 and seg["appended_postings"] == 0
    main = seg["main_postings"]
    extra = make_texts(300, seed=22, lo=5, hi=30) + ["brandnewterm w000x", "brandnewterm brandnewterm zzzunique"]
    n = len(texts)
    for i, t in enumerate(extra):
        g.add_chunk(n + i, t)
        o.add_chunk(n + i, t, rank=n + i)
        q, lim = queries[i % len(queries)]
        check(g, o, q, lim)
    check(g, o, "brandnewterm zzzunique", 10)                    # terms born after the main segment was built
    seg = g.segments()
    assert seg["full_rebuilds"] == 1 and seg["append_rebuilds"] == len(extra) and seg["main_postings"] == main
    assert seg["appended_postings"] == g.info()["n_postings"] - main > 0
    # replace a row of the appended segment: still no full rebuild
    g.add_chunk(n + 3, "w000x w000x replaced")
    o.add_chunk(n + 3, "w000x w000x replaced", rank=n + 3)
    check(g, o, "w000x replaced", 100)
    assert g.segments()["full_rebuilds"] == 1
    # replace a row of the main segment: full rebuild, appended segment folded in
    g.add_chunk(5, "w001x rewritten early row")
    o.add_chunk(5, "w001x rewritten early row", rank=5)
    check(g, o, "w001x rewritten", 100)
    seg = g.segments()
    assert seg["full_rebuilds"] == 2 and seg["appended_postings"] == 0 and seg["main_postings"] == g.info()["n_postings"]
    # removal renumbers rows: an ordered compaction of both segments on the device, no rebuild -- with appended rows present
    g.add_chunk(n + len(extra), "w000x appendedbeforeremoval")
    o.add_chunk(n + len(extra), "w000x appendedbeforeremoval", rank=n + len(extra))
    check(g, o, "appendedbeforeremoval w000x", 50)
    before = g.segments()
    assert before["appended_postings"] == 2
    g.remove_rows([0, 17, n + 1])
    o2 = OL.LexicalIndex()
    kept = [t for i, t in enumerate(texts + extra + ["w000x appendedbeforeremoval"]) if i not in (0, 17, n + 1)]
    kept[5 - 1] = "w001x rewritten early row"                   # row 5 moved down by the removal of row 0
    kept[n + 3 - 3] = "w000x w000x replaced"
    for r, t in enumerate(kept):
        o2.add_chunk(r, t, rank=r)
    for q, lim in queries + [("appendedbeforeremoval w000x", 50), ("brandnewterm zzzunique", 10)]:
        check(g, o2, q, lim)
    seg = g.segments()
    assert seg["full_rebuilds"] == 2 and seg["append_rebuilds"] == before["append_rebuilds"]   # nothing was rebuilt
    assert seg["appended_postings"] == 2 and seg["main_postings"] == g.info()["n_postings"] - 2
    # and appends keep working on the compacted segments
    g.add_chunk(len(kept), "w001x afterremoval")
    o2.add_chunk(len(kept), "w001x afterremoval", rank=len(kept))
    kept.append("w001x afterremoval")
    check(g, o2, "afterremoval w001x", 50)
    assert g.segments()["full_rebuilds"] == 2
    # an appended segment larger than max(65536, main / 8) postings is folded into the main one
    big = make_texts(3000, seed=23, lo=30, hi=60)
    for i, t in enumerate(big):
        g.add_chunk(len(kept) + i, t)
        o2.add_chunk(len(kept) + i, t, rank=len(kept) + i)
    for q, lim in queries:
        check(g, o2, q, lim)
    seg = g.segments()
    assert seg["full_rebuilds"] == 3 and seg["appended_postings"] == 0
    g.close()


def test_engine_search_text_fused_and_fallback_paths(rlr, oracle):
    """rlr_engine_search_text (RagEngine.search / search_with_diversity with the query text): BM25 beside the scan, blend,
    cut and MMR on the device.  Against the oracle's search given the oracle's BM25 pairs, over plain / stage-1 /
    diversified searches, weight overrides (w_embedding = 0 takes the host path), a top_k whose pool exceeds the fused
    kernels (host path), unknown words (embedding-only), duplicated rows (exact cosine ties) and NaN rows."""
    n, dim = 5000, 128
    texts = make_texts(n, seed=31, lo=4, hi=25)
    rows = oracle.synth_rows(n, dim, seed=32)
    rows[100:140] = rows[7]                      # exact duplicates: cosine ties across lexical and non-lexical rows
    rows[2000, 5] = np.nan                       # a NaN row scores NaN and orders last
    eng = rlr.RagEngine(dim)
    eng.add_document("d", texts, rows)
    stored = eng.index.fetch_rows(np.arange(n))
    o = OL.LexicalIndex()
    for r, t in enumerate(texts):
        o.add_chunk(r, t, rank=r)
    qs = [oracle.synth_query(dim, seed=33 + i) for i in range(3)] + [rows[7].copy()]
    cases = [
        # (text, top_k, diversity, stage, weights)
        ("w000x w001x common", 10, 0.0, 0, None),
        ("w000x w001x common", 10, 0.0, 1, None),
        ("w017x frequent", 25, 0.3, 0, None),
        ("the of w002x w003x w004x", 100, 0.7, 0, None),
        ("w005x", 7, 1.0, 0, None),
        ("nothing known here", 10, 0.5, 0, None),                                  # no lexical candidate at all
        ("", 10, 0.0, 0, None),
        ("w000x w010x", 10, 0.4, 0, dict(embedding=0.0, lexical=1.0)),             # w_e = 0: host path
        ("w000x w010x", 10, 0.0, 0, dict(embedding=1.0, lexical=0.0)),
        ("w000x w010x", 12, 0.2, 0, dict(embedding=0.25, lexical=0.75)),
        ("w001x w002x", 400, 0.5, 0, None),                                        # pool 1200 > 1024: host path
        ("w001x w002x", 0, 0.0, 0, None),                                          # top_k 0 is served as 1 (:490)
    ]
    for ci, (text, k, div, stage, wts) in enumerate(cases):
        q = qs[ci % len(qs)]
        w = rlr.QueryWeights(**wts) if wts else None
        w_e, w_l = (wts["embedding"], wts["lexical"]) if wts else (0.7, 0.3)
        lam = min(max(div, 0.0), 1.0)
        k_eff = k if lam == 0.0 else max(3 * k, k + 10)
        pairs = [(c, float(s)) for c, s in o.score(text, 5 * max(k_eff, 1), keep_zero=False)]
        if lam == 0.0:
            got = eng.search(q, k, weights=w, stage=stage, query_text=text)
            wr, wc, we, wl = oracle.search(stored, q, k, w_e, w_l, lex=pairs, stage=stage)
        else:
            got = eng.search_with_diversity(q, k, div, weights=w, query_text=text)
            wr, wc, we, wl = oracle.search_with_diversity(stored, q, k, div, w_e, w_l, lex=pairs)
        ctx = (ci, text, k, div, stage, wts)
        assert [g_.row for g_ in got] == list(wr), ctx
        assert np.array_equal(bits([g_.score for g_ in got]), bits(wc)), ctx
        assert np.array_equal(bits([g_.embedding_score for g_ in got]), bits(we)), ctx
        assert np.array_equal(bits([g_.lexical_score for g_ in got]), bits(wl)), ctx
    # the lexical index may run ahead of the embedding matrix (a chunk whose embedding is not stored yet): its rows
    # still count for max_lexical (:515-519) and never become candidates
    eng.lexical.add_chunk(n + 3, "zzzrare zzzrare zzzrare w399x")
    o.add_chunk(n + 3, "zzzrare zzzrare zzzrare w399x", rank=n + 3)
    pairs = [(c, float(s)) for c, s in o.score("zzzrare w399x", 50, keep_zero=False)]
    assert pairs[0][0] == n + 3 and len(pairs) > 1          # the chunk without an embedding holds the largest BM25 score
    got = eng.search(qs[0], 10, query_text="zzzrare w399x")
    wr, wc, we, wl = oracle.search(stored, qs[0], 10, lex=pairs)
    assert [g_.row for g_ in got] == list(wr)
    assert np.array_equal(bits([g_.score for g_ in got]), bits(wc))
    assert np.array_equal(bits([g_.lexical_score for g_ in got]), bits(wl))
    eng.close()


def test_engine_search_text_with_more_candidates_than_one_workgroup_sorts(rlr, oracle):
    """> 8192 documents hold the query's terms: BM25 goes through the sampled selection (and, in the retry test's child
    process, through its hand-back: the blend reports status 3 and the engine repeats the query on the exact path)"""
    n, dim = 12000, 64
    texts = make_texts(n, seed=41, lo=3, hi=12, common_every=1)      # every chunk holds "ubiquitous"
    rows = oracle.synth_rows(n, dim, seed=42)
    eng = rlr.RagEngine(dim)
    eng.add_document("d", texts, rows)
    stored = eng.index.fetch_rows(np.arange(n))
    o = OL.LexicalIndex()
    for r, t in enumerate(texts):
        o.add_chunk(r, t, rank=r)
    for qi, (text, k, div) in enumerate([("ubiquitous w000x w001x", 20, 0.3), ("w002x ubiquitous", 100, 0.0), ("w000x", 10, 0.7)]):
        q = oracle.synth_query(dim, seed=43 + qi)
        k_eff = k if div == 0.0 else max(3 * k, k + 10)
        pairs = [(c, float(s)) for c, s in o.score(text, 5 * k_eff, keep_zero=False)]
        got = eng.search_with_diversity(q, k, div, query_text=text)
        wr, wc, we, wl = oracle.search_with_diversity(stored, q, k, div, lex=pairs)
        assert [g_.row for g_ in got] == list(wr), (text, k, div)
        assert np.array_equal(bits([g_.score for g_ in got]), bits(wc)), (text, k, div)
        assert np.array_equal(bits([g_.lexical_score for g_ in got]), bits(wl)), (text, k, div)
        lr, ls = eng.lexical.score(text, 5 * k_eff)
        assert [int(r) for r in lr] == [c for c, _ in pairs] and np.array_equal(bits(ls), bits([s for _, s in pairs]))
    import os
    retries = eng.lexical.segments()["select_retries"]
    if os.environ.get("RLR_LEX_SAMPLE_RANK") == "1":
        assert retries >= 4, retries          # the forced short list: every large query came back through the exact path
    else:
        assert retries == 0, retries
    eng.close()


def test_engine_search_text_at_150k_chunks_sampled_selection_regime(rlr, oracle):
    """150 k chunks, common query words (n_touched ~ the whole corpus), 1500 BM25 pairs wanted: the sampled selection works
    from an 8192-entry sample of ~150 k keys (mu ~ 80 expected hits) -- the regime the C2 / 10 M-chunk deployments run
    in.  Same answer as the oracle given the oracle's pairs, and no query needed the exact-path retry."""
    n, dim = 150_000, 32
    rng = np.random.default_rng(51)
    zipf = 1.0 / np.arange(1, len(VOCAB) + 1)
    zipf /= zipf.sum()
    words = rng.choice(len(VOCAB), size=(n, 12), p=zipf)
    lens = rng.integers(3, 13, size=n)
    texts = [" ".join(VOCAB[w] for w in words[i, : lens[i]]) for i in range(n)]
    rows = oracle.synth_rows(n, dim, seed=52)
    eng = rlr.RagEngine(dim)
    eng.add_document("d", texts, rows)
    stored = eng.index.fetch_rows(np.arange(n))
    o = OL.LexicalIndex()
    for r, t in enumerate(texts):
        o.add_chunk(r, t, rank=r)
    for qi, (text, k, div) in enumerate([("w000x w001x w017x", 100, 0.3), ("w003x common", 100, 0.0), ("w100x w200x", 50, 0.7)]):
        q = oracle.synth_query(dim, seed=53 + qi)
        k_eff = k if div == 0.0 else max(3 * k, k + 10)
        pairs = [(c, float(s)) for c, s in o.score(text, 5 * k_eff, keep_zero=False)]
        got = eng.search_with_diversity(q, k, div, query_text=text)
        wr, wc, we, wl = oracle.search_with_diversity(stored, q, k, div, lex=pairs)
        assert [g_.row for g_ in got] == list(wr), (text, k, div)
        assert np.array_equal(bits([g_.score for g_ in got]), bits(wc)), (text, k, div)
        assert np.array_equal(bits([g_.lexical_score for g_ in got]), bits(wl)), (text, k, div)
    assert eng.lexical.segments()["select_retries"] == 0
    eng.close()


def test_hybrid_entry_points_reject_bad_arguments_and_stay_usable(rlr, oracle):
    """rlr_search_hybrid / rlr_engine_search_text argument checks (some of them after the scan has been enqueued: the stream is
    drained and the search context handed back), and the index answers normally afterwards"""
    import ctypes as C
    N = rlr._native
    L = N.lib()
    n, dim = 3000, 64
    rows = oracle.synth_rows(n, dim, seed=61)
    eng = rlr.RagEngine(dim)
    eng.add_document("d", make_texts(n, seed=62), rows)
    q = oracle.normalize(oracle.synth_query(dim, seed=63))
    out_r = np.zeros(64, np.uint64)
    out_f = [np.zeros(64, np.float32) for _ in range(3)]
    n_out, fb = C.c_uint32(), C.c_int32()

    def hybrid(lrows, lscores, need=10, k=5, div=1, q_=q):
        lr = np.ascontiguousarray(lrows, np.uint64)
        ls = np.ascontiguousarray(lscores, np.float32)
        return L.rlr_search_hybrid(eng.index.handle, q_.ctypes.data_as(N.f32p) if q_ is not None else None, need, k, 0.3, div, 0.7,
                                   0.3, lr.ctypes.data_as(N.u64p), ls.ctypes.data_as(N.f32p), len(lrows), 2.0, -1.0,
                                   out_r.ctypes.data_as(N.u64p), out_f[0].ctypes.data_as(N.f32p), out_f[1].ctypes.data_as(N.f32p),
                                   out_f[2].ctypes.data_as(N.f32p), C.byref(n_out), C.byref(fb))

    assert hybrid([5, 3], [1.0, 2.0]) == N.RLR_E_INVALID           # not ascending (detected after the scan was enqueued)
    assert hybrid([5, 5], [1.0, 2.0]) == N.RLR_E_INVALID           # not unique
    assert hybrid([5, n], [1.0, 2.0]) == N.RLR_E_INVALID           # outside the index
    assert hybrid([5], [1.0], q_=None) == N.RLR_E_INVALID
    assert hybrid([3, 5], [1.0, 2.0]) == N.RLR_OK and fb.value == 0 and n_out.value == 5
    assert hybrid([3, 5], [1.0, 2.0], need=2000) == N.RLR_OK and fb.value == 1   # more candidates than the fused kernels keep
    hits = (N.SearchHitC * 64)()
    st = L.rlr_engine_search_text(eng.index.handle, None, q.ctypes.data_as(N.f32p), dim, b"w000x", 5, 5, 0.3, 0, None, hits, 64,
                                  C.byref(n_out))
    assert st == N.RLR_E_INVALID                                    # no lexical index
    # everything still works
    got = eng.search_with_diversity(q, 5, 0.3, query_text="w000x w001x")
    o = OL.LexicalIndex()
    for r, t in enumerate(make_texts(n, seed=62)):
        o.add_chunk(r, t, rank=r)
    pairs = [(c, float(s)) for c, s in o.score("w000x w001x", 5 * 15, keep_zero=False)]
    wr, wc, _, _ = oracle.search_with_diversity(eng.index.fetch_rows(np.arange(n)), q, 5, 0.3, lex=pairs)
    assert [g_.row for g_ in got] == list(wr) and np.array_equal(bits([g_.score for g_ in got]), bits(wc))
    eng.close()


def clustered_texts(n, seed):
    """terms that live in row BANDS (a document ingested chunk after chunk keeps its vocabulary together): the place of a row
    in such a posting list is nowhere near cnt * row / n_rows, so the range search's estimate misses and its fallback runs"""
    rng = np.random.default_rng(seed)
    texts = []
    for r in range(n):
        words = [f"band{r * 16 // n:02d}"] * int(rng.integers(1, 4))          # 16 bands of n / 16 rows
        if r < n // 8:
            words.append("early")
        if r >= n - n // 5:
            words += ["late", "late"]
        if r % 7 == 0:
            words.append("spread")
        words += list(rng.choice(VOCAB[:60], size=int(rng.integers(2, 9))))
        texts.append(" ".join(words))
    return texts


@pytest.mark.parametrize("form", ["lds", "lds-wide", "global"])
def test_bm25_clustered_postings_both_accumulator_forms(rlr, monkeypatch, form):
    """bm25_terms_lds_kernel (sums in LDS, estimated range search with its fallback) and bm25_terms_kernel (sums in device
    memory, RLR_LEX_TERMS=global at creation) against the oracle, on posting lists clustered by row; 17 terms take two
    launches (the second continues from the first one's sums).  lds-wide: 17 workgroups of ~1900 rows each, as on an index
    of millions of rows -- a term's share of a workgroup comes in several chunks of 256 postings"""
    if form == "global":
        monkeypatch.setenv("RLR_LEX_TERMS", "global")
    if form == "lds-wide":
        monkeypatch.setenv("RLR_LEX_LDS_WGS", "17")
    n = 30000
    texts = clustered_texts(n, seed=41)
    g, o = build_pair(rlr, texts)
    many = " ".join(VOCAB[:15]) + " early late"
    for q, lim in [("early", 100), ("late spread", 1500), ("band00 band15 early late", 700), ("band07", 0),
                   ("spread w000x band03 late", 2000), (many, 1200), ("band08 band09 w001x", 5000)]:
        assert check(g, o, q, lim) > 0
    # appended rows (the second posting segment) and a removed band
    extra = clustered_texts(2000, seed=42)
    for i, t in enumerate(extra):
        g.add_chunk(n + i, t)
        o.add_chunk(n + i, t, rank=n + i)
    for q, lim in [("early late", 900), ("band15 spread", 1500)]:
        assert check(g, o, q, lim) > 0
    g.close()


@pytest.mark.parametrize("fetch", ["tight", "full"])
def test_engine_search_text_cosine_fetch_of_need_plus_a_margin_is_exact(rlr, oracle, monkeypatch, fetch):
    """The blend of a text search asks for the need + 32 best rows by cosine (rlr_index::hybrid_fetch_full; rounds 2-3:
    need + n_lexical + 8, RLR_HYBRID_FETCH=full): a lexical term only adds to a blended score, so those rows hold every
    non-lexical row that can reach the pool.  Against the oracle where the lexical term decides the pool, where the cosine
    alone does (many lexical rows among the best cosines), and with duplicated rows tying across the fetch boundary
    (status 2: the host's widening path answers)."""
    if fetch == "full":
        monkeypatch.setenv("RLR_HYBRID_FETCH", "full")
    n, dim = 20000, 64
    texts = make_texts(n, seed=51, lo=4, hi=25)
    rows = oracle.synth_rows(n, dim, seed=52)
    q_dup = oracle.synth_query(dim, seed=599)
    rows[300:420] = q_dup                      # 120 identical rows: the best cosines of q_dup tie far across need + 32
    eng = rlr.RagEngine(dim)
    eng.add_document("d", texts, rows)
    stored = eng.index.fetch_rows(np.arange(n))
    o = OL.LexicalIndex()
    for r, t in enumerate(texts):
        o.add_chunk(r, t, rank=r)
    text, k, div = "w000x w001x common the w017x", 20, 0.4
    k_eff = max(3 * k, k + 10)
    pairs = [(c, float(s)) for c, s in o.score(text, 5 * k_eff, keep_zero=False)]
    assert len(pairs) > 100

    def run(q, wts, diverse=True):
        w = rlr.QueryWeights(**wts) if wts else None
        w_e, w_l = (wts["embedding"], wts["lexical"]) if wts else (0.7, 0.3)
        if diverse:
            got = eng.search_with_diversity(q, k, div, weights=w, query_text=text)
            wr, wc, we, wl = oracle.search_with_diversity(stored, q, k, div, w_e, w_l, lex=pairs)
        else:
            got = eng.search(q, k, weights=w, query_text=text)
            wr, wc, we, wl = oracle.search(stored, q, k, w_e, w_l, lex=[(c, s) for c, s in o.score(text, 5 * k, keep_zero=False)])
        assert [g_.row for g_ in got] == list(wr), wts
        assert np.array_equal(bits([g_.score for g_ in got]), bits(wc)), wts
        assert np.array_equal(bits([g_.lexical_score for g_ in got]), bits(wl)), wts

    for i in range(6):
        q = oracle.synth_query(dim, seed=600 + i)
        run(q, None)                                            # the lexical term decides the pool
        run(q, dict(embedding=1.0, lexical=1e-6))               # the cosine decides it
        run(q, dict(embedding=0.5, lexical=0.05), diverse=False)
    # the query whose 120 best rows are identical: the need-th blended score does not beat the last fetched cosine
    run(q_dup, dict(embedding=1.0, lexical=1e-6))
    run(q_dup, None)
    eng.close()
