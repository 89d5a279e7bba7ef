"""C2 as the reference actually runs it: search_documents -> search_with_diversity with the query TEXT, so BM25
candidates (LexicalIndex::score, top_k * 5 of them) are blended into the pool before MMR.  100k x 768 f32, top_k=100,
lambda=0.3.  Times the Python engine call and its parts."""
import importlib, sys, time, json
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
rlr = importlib.import_module("rust-local-rag_amd")
n, dim, k, lam = 100_000, 768, 100, 0.3
rng = np.random.default_rng(2)
V = 20000
vocab = np.array([f"t{i:05d}" for i in range(V)])
zipf = 1.0 / np.arange(1, V + 1); zipf /= zipf.sum()
eng = rlr.RagEngine(dim)
t0 = time.perf_counter()
B = 10000
for b0 in range(0, n, B):
    emb = rng.standard_normal((B, dim)).astype(np.float32)
    words = rng.choice(V, size=(B, 40), p=zipf)
    texts = [" ".join(vocab[w]) for w in words]
    eng.add_document(f"doc{b0}", texts, emb)
print("built %d chunks in %.1f s" % (len(eng), time.perf_counter() - t0), flush=True)
qs = rng.standard_normal((300, dim)).astype(np.float32)
qtexts = [" ".join(vocab[rng.choice(V, size=6, p=zipf)]) for _ in range(300)]
def timeit(f, lo=20, hi=270):
    for i in range(lo): f(i)
    t0 = time.perf_counter()
    for i in range(lo, hi): f(i)
    return (time.perf_counter() - t0) / (hi - lo) * 1e3
import gc; gc.collect(); gc.freeze()
if "hybrid-only" in sys.argv:
    t_hyb = timeit(lambda i: eng.search_with_diversity(qs[i], k, lam, query_text=qtexts[i]))
    t_search_h = timeit(lambda i: eng.search(qs[i], k, query_text=qtexts[i]))
    print(json.dumps({"hybrid_search_with_diversity_ms": t_hyb, "search_hybrid_ms": t_search_h}))
    sys.exit(0)
t_plain = timeit(lambda i: eng.search_with_diversity(qs[i], k, lam))
t_hyb = timeit(lambda i: eng.search_with_diversity(qs[i], k, lam, query_text=qtexts[i]))
t_lex = timeit(lambda i: eng.lexical.score(qtexts[i], 5 * 300))
t_search_h = timeit(lambda i: eng.search(qs[i], k, query_text=qtexts[i]))
t_search_p = timeit(lambda i: eng.search(qs[i], k))
r = eng.search_with_diversity(qs[0], k, lam, query_text=qtexts[0])
print(json.dumps({"search_with_diversity_ms": t_plain, "hybrid_search_with_diversity_ms": t_hyb, "lexical_score_alone_ms": t_lex,
                  "search_plain_ms": t_search_p, "search_hybrid_ms": t_search_h,
                  "n_results": len(r), "n_with_lexical": sum(1 for x in r if x.lexical_score)}))
