"""f4 on the GPU: the reference harness' view (HTTP /search -> keys -> Hit@k / MRR / NDCG) is the same
whether the results come from the GPU path or from the oracle's CPU restatement of the same search."""
import importlib

import numpy as np
import pytest

from oracle import lexical as OL
from test_gpu_lexical import make_texts

pytestmark = pytest.mark.gpu


def test_product_level_no_regression_over_http(rlr, oracle):
    ek = importlib.import_module("rust-local-rag_amd.evalkit")
    dim, n_docs, per_doc = 256, 12, 60
    n = n_docs * per_doc
    rows = oracle.synth_rows(n, dim, seed=71, n_clusters=30)
    texts = make_texts(n, seed=72)
    eng = rlr.RagEngine(dim)
    for d in range(n_docs):
        sl = slice(d * per_doc, (d + 1) * per_doc)
        eng.add_document(f"Book {d}.pdf", texts[sl], rows[sl], pages=[1 + i // 3 for i in range(per_doc)])
    stored = eng.index.fetch_rows(np.arange(n))
    o = OL.LexicalIndex()
    for r, t in enumerate(texts):
        o.add_chunk(r, t, rank=r)

    # queries: a noisy copy of a chunk's embedding + a few of its words; gold = that chunk's document/page
    rng = np.random.default_rng(73)
    emb_of, queries = {}, []
    for i in range(24):
        r = int(rng.integers(0, n))
        words = texts[r].replace(",", " ").replace("-", " ").split()
        text = " ".join(words[:3]) + f" q{i}"
        emb_of[text] = (rows[r] + 0.35 * rng.standard_normal(dim).astype(np.float32) / np.sqrt(dim)).astype(np.float32)
        ch = eng._chunks[r]
        queries.append({"query_id": f"Q{i}", "query": text,
                        "gold_references": [{"document": ch.document_name, "page": ch.page_number, "relevance": 3}]})
    queries.append({"query_id": "QR", "query": "zzz", "is_rejection": True, "gold_references": []})
    emb_of["zzz"] = rng.standard_normal(dim).astype(np.float32)

    server, _ = ek.serve(ek.SearchService(eng, embed=lambda t: emb_of[t]))
    try:
        gpu_search = ek.http_search_fn("http://127.0.0.1:%d" % server.server_address[1])

        def cpu_search(text, top_k):  # the oracle's search_with_diversity with the oracle's BM25 pairs
            k_eff = max(3 * top_k, top_k + 10)
            pairs = [(c, float(s)) for c, s in o.score(text, 5 * k_eff, keep_zero=False)]
            wr, wc, _, _ = oracle.search_with_diversity(stored, emb_of[text], top_k, 0.3, lex=pairs)
            return [ek.RetrievedChunk(eng._chunks[int(r)].id, eng._chunks[int(r)].document_name,
                                      eng._chunks[int(r)].page_number, eng._chunks[int(r)].text, float(s))
                    for r, s in zip(wr, wc)]

        g_scores, g_sum = ek.evaluate(queries, gpu_search, k=5)
        c_scores, c_sum = ek.evaluate(queries, cpu_search, k=5)
        for a, b in zip(g_scores, c_scores):
            assert a.retrieved_keys == b.retrieved_keys, a.query_id
            assert (a.hit_rate, a.mrr, a.ndcg, a.precision) == (b.hit_rate, b.mrr, b.ndcg, b.precision)
        for key in ("hit_rate_mean", "mrr_mean", "ndcg_mean", "precision_mean"):
            assert g_sum[key] == c_sum[key]
        assert g_sum["hit_rate_mean"] > 0.8          # the synthetic task is easy: the harness does find the gold chunks
        # scores over the wire are the f32 values the oracle computed
        got = gpu_search(queries[0]["query"], 5)
        want = cpu_search(queries[0]["query"], 5)
        assert [np.float32(r.score) for r in got] == [np.float32(r.score) for r in want]
    finally:
        server.shutdown()
        server.server_close()
        eng.close()
