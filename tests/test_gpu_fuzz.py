"""Randomised differential test on the GPU: search_topk (single queries, small and large batches, with and without
the nomination image / image scan, f32 and f16 rows, every row-pitch class, duplicate / zero / NaN rows) and MMR
against the oracle.  Found the sign-of-zero difference in the logged MMR value at lambda = 1 (fixed in exact.hip).
Every fuzzer runs a fixed NUMBER of cases from its seed (not a wall-clock budget), so the case set is the same on
every box.  Standalone: python tests/test_gpu_fuzz.py <cases> [seed]."""
import importlib
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DIMS = [64, 100, 128, 192, 256, 320, 384, 512, 640, 768, 896, 1000, 1024, 1152, 1280, 1536, 2048]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def fuzz(n_target: int, seed0: int):
    rlr = importlib.import_module("rust-local-rag_amd")
    from oracle import oracle as O

    def otopk(rows, qn, k):
        e = O.scan(rows, qn)
        key = np.where(np.isnan(e), -np.inf, e)
        order = np.lexsort((np.arange(len(e)), -key.astype(np.float64)))[:k]
        return order.astype(np.uint64), e[order]

    rng = np.random.default_rng(seed0)
    n_cases = n_q = 0
    while n_cases < n_target:
        dim = int(rng.choice(DIMS)); dtype = str(rng.choice(["f32", "f32", "f16"]))
        n = int(rng.choice([1, 2, 17, 63, 64, 65, 255, 256, 257, 1000, 4095, 4096, 4097, 9000, 20011]))
        ncl = int(rng.choice([0, 0, 3, 50]))
        seed = int(rng.integers(1, 1 << 30))
        rows = O.synth_rows(n, dim, seed=seed, n_clusters=ncl, f16=(dtype == "f16"))
        # special rows: duplicates, zeros, a NaN row, a scaled row
        if n > 20 and rng.random() < 0.5:
            rows[5] = rows[3]; rows[n - 1] = rows[3]; rows[7] = 0
            if dtype == "f32" and rng.random() < 0.5: rows[9, 0] = np.nan
        ix = rlr.GpuIndex(dim, dtype); ix.upload(rows)
        mode = str(rng.choice(["plain", "image", "image_scan", "q8"])) if dim % 64 == 0 else "plain"
        if mode == "q8":
            if dim <= 2048 and dim % 16 == 0: ix.enable_batch_image(False, q8=True)
        elif mode != "plain": ix.enable_batch_image(True, single_query=(mode == "image_scan"))
        nq = int(rng.choice([1, 1, 2, 5, 20, 40]))
        k = int(min(rng.choice([1, 5, 10, 100, 300, 1000]), max(n, 1) + 3))
        qs = np.stack([O.normalize(O.synth_query(dim, seed=seed + 7 + i)) for i in range(nq)])
        if rng.random() < 0.3 and n > 3: qs[0] = O.normalize(rows[3].copy()) if np.isfinite(rows[3]).all() and rows[3].any() else qs[0]
        r, c = ix.search_topk(qs, k)
        for i in range(nq):
            wr, wc = otopk(rows, qs[i], k)
            ok = np.array_equal(r[i], wr) and np.array_equal(bits(c[i]), bits(wc))
            if not ok:
                print("MISMATCH", dict(dim=dim, dtype=dtype, n=n, ncl=ncl, seed=seed, mode=mode, nq=nq, k=k, q=i)); raise AssertionError("see the MISMATCH line above")
            n_q += 1
        # MMR on the first query's pool
        if n >= 3 and np.isfinite(c[0]).all():
            P = min(len(r[0]), 60); kk = int(rng.integers(1, P + 1)); lam = float(rng.choice([0.0, 0.3, 0.7, 1.0]))
            sc = (np.float32(0.7) * c[0][:P]).astype(np.float32)
            o, m = ix.mmr_select(r[0][:P], sc, kk, lam)
            wo, wm = O.mmr(rows[r[0][:P].astype(np.int64)], sc, kk, lam)
            if not (np.array_equal(o, wo) and np.array_equal(bits(m[1:]), bits(wm[1:]))):
                print("MMR MISMATCH", dict(dim=dim, dtype=dtype, n=n, seed=seed, P=P, kk=kk, lam=lam)); raise AssertionError("see the MISMATCH line above")
        ix.close(); n_cases += 1
        if n_cases % 100 == 0 and os.environ.get("RLR_FUZZ_PROGRESS"):
            print("fuzz: %d corpora" % n_cases, flush=True)
    return n_cases, n_q


def fuzz_engine(n_target: int, seed0: int):
    """RagEngine level: documents added / replaced / removed, hybrid search from caller pairs and from query text
    (GPU BM25), weight overrides, stage-1 candidates + reranker blend, search_with_diversity -- against the oracle."""
    rlr = importlib.import_module("rust-local-rag_amd")
    from oracle import oracle as O
    from oracle import lexical as OL

    rng = np.random.default_rng(seed0)
    vocab = [f"t{i:03d}w" for i in range(120)]
    n_cases = 0
    while n_cases < n_target:
        dim = int(rng.choice([64, 256, 384, 768]))
        eng = rlr.RagEngine(dim, accelerate=[None, None, "image", "q8"][int(rng.integers(0, 4))])
        texts_of, n_docs = {}, int(rng.integers(1, 6))
        for d in range(n_docs):
            m = int(rng.choice([1, 7, 60, 400]))
            texts_of[f"doc{d}.pdf"] = [" ".join(rng.choice(vocab, size=int(rng.integers(1, 25)))) for _ in range(m)]
            eng.add_document(f"doc{d}.pdf", texts_of[f"doc{d}.pdf"],
                             O.synth_rows(m, dim, seed=int(rng.integers(1, 1 << 30)), n_clusters=int(rng.choice([0, 4]))))
        if n_docs > 1 and rng.random() < 0.5:     # drop one, replace another
            eng.remove_document("doc0.pdf")
            texts_of.pop("doc0.pdf")
            m = int(rng.integers(1, 50))
            texts_of["doc1.pdf"] = [" ".join(rng.choice(vocab, size=5)) for _ in range(m)]
            eng.add_document("doc1.pdf", texts_of["doc1.pdf"], O.synth_rows(m, dim, seed=int(rng.integers(1, 1 << 30))))
        n = len(eng)
        stored = eng.index.fetch_rows(np.arange(n))
        all_texts = [ch.text for ch in eng._chunks]
        olex = OL.LexicalIndex()
        for r, t in enumerate(all_texts):
            olex.add_chunk(r, t, rank=r)
        for _ in range(4):
            q = O.synth_query(dim, seed=int(rng.integers(1, 1 << 30)))
            k = int(rng.choice([1, 3, 5, 10, 50, 100]))
            div = float(rng.choice([0.0, 0.0, 0.3, 0.7, 1.0]))
            we, wl = (0.7, 0.3)
            w = None
            if rng.random() < 0.4:
                we, wl = float(rng.choice([0.0, 0.2, 1.0])), float(rng.choice([0.0, 0.5, 0.9]))
                w = rlr.QueryWeights(embedding=we, lexical=wl)
            k_eff = k if div == 0.0 else max(3 * k, k + 10)
            if rng.random() < 0.5:               # GPU BM25 from the query text
                text = " ".join(rng.choice(vocab, size=int(rng.integers(1, 5))))
                pairs = [(int(c), float(sc)) for c, sc in olex.score(text, 5 * k_eff, keep_zero=False)]
                got = eng.search_with_diversity(q, k, div, weights=w, query_text=text)
            else:                                 # caller-supplied pairs, ties and zeros included
                m = int(rng.integers(0, min(n, 12) + 1))
                rows_l = rng.choice(n, size=m, replace=False)
                pairs = [(int(r), float(rng.choice([0.0, 0.5, 2.25, 7.0]))) for r in rows_l]
                got = eng.search_with_diversity(q, k, div, weights=w, lexical=[(eng._chunks[r].id, sc) for r, sc in pairs])
            wr, wc, we_o, wl_o = O.search_with_diversity(stored, q, k, div, w_e=we, w_l=wl, lex=pairs)
            ctx = dict(dim=dim, n=n, k=k, div=div, we=we, wl=wl, pairs=len(pairs), seed0=seed0, case=n_cases)
            assert [g.row for g in got] == list(wr), ctx
            assert np.array_equal(bits([g.score for g in got]), bits(wc)), ctx
            assert np.array_equal(bits([g.lexical_score for g in got]), bits(wl_o)), ctx
        # stage-1 candidates (what a reranker receives) + the blend with a random reranker answer
        q = O.synth_query(dim, seed=int(rng.integers(1, 1 << 30)))
        k = int(rng.choice([1, 5, 20]))
        cands = eng.search(q, k, stage=1)
        wr, wc, _, _ = O.search(stored, q, k, stage=1)
        assert [g.row for g in cands] == list(wr) and np.array_equal(bits([g.score for g in cands]), bits(wc)), ("stage1", seed0, n_cases)
        m = int(rng.integers(0, len(cands) + 1))
        picks = rng.permutation(len(cands))[:m]
        rel = rng.random(m).astype(np.float32)
        final = eng.finish_with_reranker(cands, [(cands[int(p_)].chunk_id, float(rel[j])) for j, p_ in enumerate(picks)], k)
        bi, bs, _, _ = O.blend(wr, wc, [int(wr[int(p_)]) for p_ in picks], rel, k)
        assert [g.row for g in final] == [int(wr[int(i)]) for i in bi], ("blend", seed0, n_cases)
        assert np.array_equal(bits([g.score for g in final]), bits(bs)), ("blend scores", seed0, n_cases)
        # get_embedding_candidates and the batched diversity entry point
        cnt = int(rng.choice([1, 10, 200]))
        ec = eng.get_embedding_candidates(q, cnt)
        er, es = O.embedding_candidates(stored, q, cnt)
        assert [eng._row_of[c] for c, _ in ec] == list(er) and np.array_equal(bits([x for _, x in ec]), bits(es)), ("ec", seed0, n_cases)
        nqb = int(rng.choice([2, 5, 30]))
        qb = np.stack([O.synth_query(dim, seed=int(rng.integers(1, 1 << 30))) for _ in range(nqb)])
        kb, divb = int(rng.choice([3, 10])), float(rng.choice([0.0, 0.5]))
        got_b = eng.search_with_diversity_batch(qb, kb, divb)
        for i in range(nqb):
            wr, wc, _, _ = O.search_with_diversity(stored, qb[i], kb, divb)
            assert [g.row for g in got_b[i]] == list(wr) and np.array_equal(bits([g.score for g in got_b[i]]), bits(wc)), ("batch div", seed0, n_cases, i)
        eng.close()
        n_cases += 1
    return n_cases


def fuzz_multi(n_target: int, seed0: int):
    """in-process multi-shard index (several shards on one GPU) against the single index: search (single and
    batched), row scoring / fetching, MMR across shards"""
    rlr = importlib.import_module("rust-local-rag_amd")
    from oracle import oracle as O

    rng = np.random.default_rng(seed0)
    n_cases = 0
    while n_cases < n_target:
        dim = int(rng.choice([64, 384, 768, 1024]))
        dtype = str(rng.choice(["f32", "f16"]))
        n = int(rng.choice([1, 5, 77, 1000, 12000]))
        shards = int(rng.choice([1, 2, 3, 5]))
        rows = O.synth_rows(n, dim, seed=int(rng.integers(1, 1 << 30)), n_clusters=int(rng.choice([0, 6])), f16=(dtype == "f16"))
        if n > 10:
            rows[n - 1] = rows[0]                                   # a cross-shard exact tie
        one = rlr.GpuIndex(dim, dtype)
        one.upload(rows)
        mi = rlr.MultiGpuIndex(dim, [0] * shards, dtype)
        mi.upload(rows)
        nq = int(rng.choice([1, 3, 20]))
        k = int(rng.choice([1, 10, 100]))
        qs = np.stack([O.normalize(O.synth_query(dim, seed=int(rng.integers(1, 1 << 30)))) for _ in range(nq)])
        if n > 10:
            qs[0] = O.normalize(rows[0].copy())
        r1, c1 = one.search_topk(qs, k)
        rm, cm = mi.search_topk(qs, k)
        ctx = dict(dim=dim, dtype=dtype, n=n, shards=shards, nq=nq, k=k, seed0=seed0, case=n_cases)
        for i in range(nq):                                         # both against the oracle: a failure names the culprit
            sc_all = O.scan(rows, qs[i])
            order = np.argsort(-sc_all, kind="stable")[:k]
            for name, (rr, cc) in (("single", (r1, c1)), ("multi", (rm, cm))):
                assert np.array_equal(rr[i], order.astype(np.uint64)) and np.array_equal(bits(cc[i]), bits(sc_all[order])), (name, i, ctx)
        assert np.array_equal(r1, rm) and np.array_equal(bits(c1), bits(cm)), ctx
        pick = rng.choice(n, size=min(n, 9), replace=False).astype(np.uint64)
        assert np.array_equal(bits(one.score_rows(qs[0], pick)), bits(mi.score_rows(qs[0], pick))), ctx
        assert np.array_equal(one.fetch_rows(pick).view(np.uint32), mi.fetch_rows(pick).view(np.uint32)), ctx
        P = min(r1.shape[1], 40)
        if P >= 2:
            sc = (np.float32(0.7) * c1[0][:P]).astype(np.float32)
            kk, lam = int(rng.integers(1, P + 1)), float(rng.choice([0.0, 0.4, 1.0]))
            o1, m1 = one.mmr_select(r1[0][:P], sc, kk, lam)
            om, mm = mi.mmr_select(r1[0][:P], sc, kk, lam)
            assert np.array_equal(o1, om) and np.array_equal(bits(m1[1:]), bits(mm[1:])), ("mmr", ctx)
        # the engine-level entry points over the shards (rlr_multi_engine_*) against the oracle's search over the whole corpus
        k2, lam2 = int(rng.choice([0, 1, 7, 40])), float(rng.choice([0.0, 0.3, 0.7, 1.0]))
        n_lex = int(rng.choice([0, 0, 3, 25]))
        lex = [(int(r), float(np.float32(x))) for r, x in zip(rng.integers(0, n + 2, size=n_lex), rng.random(n_lex) * 5)]
        lr, ls = [r for r, _ in lex], [x for _, x in lex]
        w_e = float(rng.choice([0.7, 0.7, 1.0, 0.0, 1e-40]))
        w = rlr.QueryWeights(embedding=w_e)
        for i in range(min(nq, 2)):
            wr, wc, we, wl = O.search_with_diversity(rows, qs[i], k2, lam2, w_e=w_e, lex=lex)
            h = mi.engine_search_with_diversity(qs[i], k2, lam2, weights=w, lex_rows=lr, lex_scores=ls)
            assert [int(r) for r in h["row"]] == [int(r) for r in wr] and np.array_equal(bits(h["score"]), bits(wc)) and \
                np.array_equal(bits(h["embedding_score"]), bits(we)) and np.array_equal(bits(h["lexical_score"]), bits(wl)), ("engine", i, k2, lam2, w_e, n_lex, ctx)
            wr, wc, we, wl = O.search(rows, qs[i], k2, w_e=w_e, lex=lex, stage=1)
            h = mi.engine_search(qs[i], k2, weights=w, lex_rows=lr, lex_scores=ls, stage=1)
            assert [int(r) for r in h["row"]] == [int(r) for r in wr] and np.array_equal(bits(h["score"]), bits(wc)), ("engine stage 1", i, k2, w_e, n_lex, ctx)
        hb = mi.engine_search_with_diversity_batch(qs, k2, lam2, weights=w)
        for i in range(nq):
            wr, wc, we, _ = O.search_with_diversity(rows, qs[i], k2, lam2, w_e=w_e)
            assert [int(r) for r in hb[i]["row"]] == [int(r) for r in wr] and np.array_equal(bits(hb[i]["score"]), bits(wc)) and \
                np.array_equal(bits(hb[i]["embedding_score"]), bits(we)), ("engine batch", i, k2, lam2, w_e, ctx)
        one.close()
        mi.close()
        n_cases += 1
    return n_cases


def fuzz_mmr(n_target: int, seed0: int):
    """MMR alone, aimed at the greedy kernel's tie-break and at the Gram kernels: pools of 1..1024 candidates drawn (with heavy
    repetition) from a handful of distinct rows, so similarities and MMR values tie by the dozen; relevance from a small set of
    values incl. zeros of both signs, non-finite and huge ones; every lambda regime; f32 and binary16 rows, row lengths that are
    and are not multiples of 4; single pools and batches of pools (ragged sizes) -- picks and logged values bit-equal to the
    oracle's literal loop."""
    rlr = importlib.import_module("rust-local-rag_amd")
    from oracle import oracle as O

    rng = np.random.default_rng(seed0)
    n_cases = 0
    while n_cases < n_target:
        dim = int(rng.choice([8, 32, 96, 100, 128, 257])); f16 = bool(rng.random() < 0.3) and dim % 8 == 0
        n = int(rng.choice([40, 300, 1500]))
        seed = int(rng.integers(1, 1 << 30))
        rows = O.synth_rows(n, dim, seed=seed, n_clusters=int(rng.choice([0, 2, 9])), f16=f16)
        distinct = int(rng.choice([1, 3, 17, n]))            # how many different rows the corpus really has
        if distinct < n:
            rows = rows[rng.integers(0, distinct, size=n)].copy()
        if rng.random() < 0.3:
            rows[rng.integers(0, n, 2)] = 0                  # zero rows: every similarity +0 or -0
        if not f16 and rng.random() < 0.2:
            rows[rng.integers(0, n), rng.integers(0, dim)] = np.inf   # non-finite similarities
            rows[rng.integers(0, n), rng.integers(0, dim)] = np.nan
        ix = rlr.GpuIndex(dim, "f16" if f16 else "f32"); ix.upload(rows)
        stored = ix.fetch_rows(np.arange(n))
        for _ in range(4):
            P = int(rng.choice([1, 2, 3, 63, 64, 65, 128, 129, 300, 320, 321, 512, 513, 1000, 1024]))
            pool = rng.integers(0, n, size=P).astype(np.uint64)      # (repeats allowed: exact duplicate candidates)
            levels = np.array([0.0, -0.0, 0.25, 0.5, 0.5, 1.0, -1.0, 3.0e38, -3.0e38, 1e-40], np.float32)
            sc = levels[rng.integers(0, int(rng.choice([2, 4, len(levels)])), size=P)].astype(np.float32)
            if rng.random() < 0.5:
                sc = (rng.standard_normal(P) * 0.3).astype(np.float32)
                sc[rng.integers(0, P, max(P // 8, 1))] = sc[0]
            if rng.random() < 0.3 and P > 3:
                sc[rng.integers(1, P, 3)] = [np.nan, np.inf, -np.inf]
            lam = float(rng.choice([0.0, 0.3, 0.5, 0.7, 1.0]))
            kk = int(rng.choice([0, 1, 2, min(P, 100), P, P + 5]))
            o, m = ix.mmr_select(pool, sc, kk, lam)
            wo, wm = O.mmr(stored[pool.astype(np.int64)], sc, kk, lam)
            if not (np.array_equal(o, wo) and np.array_equal(bits(m[1:]), bits(wm[1:]))):
                print("MMR MISMATCH", dict(dim=dim, f16=f16, n=n, seed=seed, distinct=distinct, P=P, kk=kk, lam=lam, case=n_cases)); raise AssertionError("see the MISMATCH line above")
        # a batch of pools with ragged sizes (the batched Gram kernels; binary16 rows from 16 pools up take the f32 matrix cores)
        m_pools = int(rng.choice([3, 9, 17])); Pb = int(rng.choice([5, 64, 130, 308]))
        prow = rng.integers(0, n, size=(m_pools, Pb)).astype(np.uint64)
        psc = (rng.integers(0, 5, size=(m_pools, Pb)) * np.float32(0.25)).astype(np.float32)
        sizes = rng.integers(0, Pb + 1, size=m_pools).astype(np.uint32); sizes[0] = Pb
        lam = float(rng.choice([0.0, 0.4, 1.0])); kk = int(rng.choice([1, 20, Pb]))
        ob, mb, nb = ix.mmr_select_batch(prow, psc, sizes, kk, lam)
        for q in range(m_pools):
            sz = int(sizes[q])
            wo, wm = O.mmr(stored[prow[q, :sz].astype(np.int64)], psc[q, :sz], kk, lam) if sz else (np.zeros(0, np.uint32), np.zeros(0, np.float32))
            if not (int(nb[q]) == len(wo) and np.array_equal(ob[q, :nb[q]], wo) and np.array_equal(bits(mb[q, 1:nb[q]]), bits(wm[1:]))):
                print("BATCH MMR MISMATCH", dict(dim=dim, f16=f16, n=n, seed=seed, distinct=distinct, Pb=Pb, q=q, sz=sz, kk=kk, lam=lam, case=n_cases)); raise AssertionError("see the MISMATCH line above")
        ix.close(); n_cases += 1
        if n_cases % 50 == 0 and os.environ.get("RLR_FUZZ_PROGRESS"):
            print("mmr fuzz: %d corpora" % n_cases, flush=True)   # (a long run that prints nothing looks hung)
    return n_cases


def test_fuzz_mmr_ties_and_awkward_pools():
    assert fuzz_mmr(40, 77001) == 40


def test_fuzz_multi_shard_index():
    assert fuzz_multi(30, 808) == 30


def test_fuzz_engine_against_the_oracle():
    assert fuzz_engine(40, 4242) == 40


def fuzz_lexical(n_target: int, seed0: int):
    """GPU BM25 alone: random vocabularies (ASCII and not), add / replace / remove sequences, random limits."""
    lex = importlib.import_module("rust-local-rag_amd.lexical")
    from oracle import lexical as OL

    rng = np.random.default_rng(seed0)
    n_cases = 0
    while n_cases < n_target:
        V = int(rng.choice([5, 40, 400]))
        vocab = [f"w{i}q" for i in range(V)] + ["Straße", "ÉTÉ", "naïve", "日本語テキスト", "ab", "x"]
        n = int(rng.choice([1, 3, 50, 700, 9000]))
        texts = [" ".join(rng.choice(vocab, size=int(rng.integers(0, 12)))) + rng.choice(["", ".", " --", "!?"]) for _ in range(n)]
        g = lex.LexicalIndex(0)
        for r, t in enumerate(texts):
            g.add_chunk(r, t)
        cur = list(texts)
        for _ in range(int(rng.integers(0, 4))):           # mutations
            if rng.random() < 0.5 and len(cur) > 1:
                dead = sorted(set(int(x) for x in rng.choice(len(cur), size=int(rng.integers(1, min(len(cur), 6))), replace=False)))
                g.remove_rows(dead)
                cur = [t for r, t in enumerate(cur) if r not in set(dead)]
            else:
                r = int(rng.integers(0, len(cur) + 1))
                t = " ".join(rng.choice(vocab, size=int(rng.integers(0, 8))))
                g.add_chunk(r, t)
                if r == len(cur):
                    cur.append(t)
                else:
                    cur[r] = t
        o = OL.LexicalIndex()
        for r, t in enumerate(cur):
            o.add_chunk(r, t, rank=r)
        info = g.info()
        assert info["total_docs"] == o.total_docs and info["total_length"] == o.total_length, (seed0, n_cases)
        for _ in range(5):
            q = " ".join(rng.choice(vocab, size=int(rng.integers(0, 6))))
            lim = int(rng.choice([0, 1, 5, 100, 1500, 8192]))
            rows, sc = g.score(q, lim)
            want = o.score(q, lim, keep_zero=False)
            if lim == 0:
                want = want[:8192]
            ctx = dict(seed0=seed0, case=n_cases, q=q, lim=lim, n=len(cur))
            assert [int(r) for r in rows] == [c for c, _ in want], ctx
            assert np.array_equal(bits(sc), bits([x for _, x in want])), ctx
        g.close()
        n_cases += 1
    return n_cases


def test_fuzz_lexical_against_the_oracle():
    assert fuzz_lexical(30, 555) == 30


def fuzz_scale(n_target: int, seed0: int):
    """Corpora too large for the oracle (generated on the device): every batch shape x nomination mode must give
    the same rows and score bits as the plain single-query f32 pipeline, whose own results are checked through
    size-independent properties (emitted score == reference-order re-score, nothing in a random sample beats the
    k-th result)."""
    rlr = importlib.import_module("rust-local-rag_amd")
    rng = np.random.default_rng(seed0)
    n_cases = 0
    while n_cases < n_target:
        dim = int(rng.choice([256, 768, 1024, 1152]))
        dtype = str(rng.choice(["f32", "f32", "f16"]))
        n = int(rng.choice([5000, 70_000, 300_001, 1_500_000]))
        k = int(rng.choice([1, 10, 100, 300]))
        ix = rlr.GpuIndex(dim, dtype)
        ix.fill_synthetic(n, seed=int(rng.integers(1, 1 << 30)), n_clusters=int(rng.choice([0, 64])))
        nq = int(rng.choice([2, 8, 40, 64, 130, 300]))
        qs = np.stack([rlr.normalize(rng.standard_normal(dim).astype(np.float32)) for _ in range(nq)])
        base = [ix.search_topk(qs[i], k) for i in range(min(nq, 4))]           # plain single-query pipeline
        for i, (r, c) in enumerate(base):
            assert np.array_equal(bits(c[0]), bits(ix.score_rows(qs[i], r[0]))), ("rescore", dim, dtype, n, k)
            sample = rng.choice(n, size=min(n, 20000), replace=False).astype(np.uint64)
            sc = ix.score_rows(qs[i], sample)
            assert sc[~np.isin(sample, r[0])].max(initial=-2.0) <= c[0][-1], ("topk", dim, dtype, n, k)
        for mode in ("plain", "image", "image_scan", "q8"):
            if mode == "q8":
                if dim > 2048:
                    continue
                ix.enable_batch_image(True, q8=True)
            elif mode != "plain":
                ix.enable_batch_image(True, single_query=(mode == "image_scan"))
            rb, cb = ix.search_topk(qs, k)
            for i, (r, c) in enumerate(base):
                ctx = dict(mode=mode, dim=dim, dtype=dtype, n=n, k=k, nq=nq, q=i, seed0=seed0, case=n_cases)
                assert np.array_equal(rb[i], r[0]) and np.array_equal(bits(cb[i]), bits(c[0])), ctx
            r1, c1 = ix.search_topk(qs[0], k)
            assert np.array_equal(r1[0], base[0][0][0]) and np.array_equal(bits(c1[0]), bits(base[0][1][0])), (mode, "single")
        ix.close()
        n_cases += 1
    return n_cases


def test_fuzz_batch_shapes_and_nomination_modes_at_scale():
    assert fuzz_scale(5, 31337) == 5


def test_fuzz_against_the_oracle():
    n_cases, n_q = fuzz(200, 20261004)
    assert n_cases == 200 and n_q >= 200


def test_poisoned_allocations():
    """the same differential tests in a child process whose every device allocation starts out as 0xFF bytes
    (RLR_POISON_ALLOC=1: NaN rows, huge counters) -- a kernel that reads memory nobody wrote cannot hide behind the
    zero pages of a fresh process"""
    import os
    import subprocess

    env = dict(os.environ, RLR_POISON_ALLOC="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-u", os.path.abspath(__file__), "100", "9"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "multi-shard fuzz ok" in out.stdout
    # and the directed tests of the paths with the most scratch buffers: large-candidate finishes, fused / hybrid searches,
    # the lexical index (a large-candidate finish once sorted the uninitialised tail of its key buffer: only this mode saw it)
    sel = "k_larger or dense or overflow or flood or search_diverse or config2 or hybrid or text or bm25 or mmr_single"
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k", sel,
                          os.path.join("tests", "test_gpu_parity.py"), os.path.join("tests", "test_gpu_lexical.py")],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


def test_lexical_sampled_selection_retry_path():
    """RLR_LEX_SAMPLE_RANK=1 in a child process: the sampled BM25 selection takes the largest sample key as threshold, its
    candidate list comes out short, and every large query goes through the retry -- host API (rlr_lexical_score repeats
    the query on the exact radix path) and fused hybrid search (the blend hands the query back, status 3) alike.  Same
    differential tests, same answers."""
    import os
    import subprocess

    env = dict(os.environ, RLR_LEX_SAMPLE_RANK="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, 'tests'); import test_gpu_fuzz as F; "
            "print('lexical fuzz ok: %d' % F.fuzz_lexical(12, 77)); print('engine fuzz ok: %d' % F.fuzz_engine(6, 78)); "
            "import test_gpu_lexical as L, importlib, conftest; rlr = conftest.load_pkg(); from oracle import oracle as O; O.lib(); "
            "L.test_bm25_select_path_many_postings_and_ties(rlr); L.test_engine_search_text_fused_and_fallback_paths(rlr, O); L.test_engine_search_text_with_more_candidates_than_one_workgroup_sorts(rlr, O); "
            "print('retry path ok')")
    out = subprocess.run([sys.executable, "-u", "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "retry path ok" in out.stdout


def test_mmr_logged_value_keeps_the_sign_of_zero(rlr, oracle):
    """lambda = 1: (1 - lambda) * rel is -0.0 for a negative relevance and the reference logs -0.0 - 0.0 = -0.0"""
    rows = oracle.synth_rows(17, 1152, seed=470119562, f16=True)
    ix = rlr.GpuIndex(1152, "f16")
    ix.upload(rows)
    q = oracle.normalize(oracle.synth_query(1152, seed=470119562 + 7))
    r, c = ix.search_topk(q, 17)
    sc = (np.float32(0.7) * c[0]).astype(np.float32)
    o, m = ix.mmr_select(r[0], sc, 17, 1.0)
    wo, wm = oracle.mmr(rows[r[0].astype(np.int64)], sc, 17, 1.0)
    assert np.array_equal(o, wo) and np.array_equal(bits(m[1:]), bits(wm[1:]))
    assert np.signbit(wm[1]) and wm[1] == 0.0          # the case that used to differ
    ob, mb, nb = ix.mmr_select_batch(r, sc[None, :], np.array([17], np.uint32), 17, 1.0)
    assert np.array_equal(ob[0, :17], wo) and np.array_equal(bits(mb[0, 1:17]), bits(wm[1:]))
    ix.close()


if __name__ == "__main__":
    sys.path.insert(0, ".")
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if len(sys.argv) > 3 and sys.argv[3] == "mmr":
        print("mmr fuzz ok: %d corpora" % fuzz_mmr(cases, seed))
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "multi":
        print("multi-shard fuzz ok: %d corpora" % fuzz_multi(cases, seed))
        sys.exit(0)
    print("fuzz ok: %d corpora, %d queries" % fuzz(cases, seed))
    print("engine fuzz ok: %d engines" % fuzz_engine(max(cases // 8, 1), seed))
    print("lexical fuzz ok: %d indexes" % fuzz_lexical(max(cases // 8, 1), seed))
    print("scale fuzz ok: %d corpora" % fuzz_scale(max(cases // 50, 1), seed))
    print("multi-shard fuzz ok: %d corpora" % fuzz_multi(max(cases // 8, 1), seed))
