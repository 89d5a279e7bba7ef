"""BM25 top-`limit` at scale: the sampled selection (limit <= 4096) against the exact eight-pass select (limit 5000 takes it),
same index, same queries -- the first `limit` of the exact answer must be the sampled answer, bit for bit.
python scratch/scale_lexical_check.py <docs>"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
lex = importlib.import_module("rust-local-rag_amd.lexical")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rng = np.random.default_rng(3)
V = 30000
vocab = np.array([f"t{i:05d}" for i in range(V)])
zipf = 1.0 / np.arange(1, V + 1); zipf /= zipf.sum()
g = lex.LexicalIndex(0)
t0 = time.perf_counter()
B = 20000
for b0 in range(0, n, B):
    m = min(B, n - b0)
    words = rng.choice(V, size=(m, 24), p=zipf)
    lens = rng.integers(8, 25, size=m)
    for i in range(m):
        g.add_tokens(b0 + i, vocab[words[i, : lens[i]]])
print("built %d docs in %.1f s %s" % (n, time.perf_counter() - t0, g.info()), flush=True)
bad = 0
for qi in range(12):
    toks = list(vocab[rng.choice(V, size=int(rng.integers(1, 6)), p=zipf)])
    for limit in (100, 500, 1500, 4000):
        t0 = time.perf_counter(); r1, s1 = g.score_tokens(toks, limit); t1 = time.perf_counter() - t0
        t0 = time.perf_counter(); r2, s2 = g.score_tokens(toks, 5000); t2 = time.perf_counter() - t0
        ok = np.array_equal(r1, r2[: len(r1)]) and np.array_equal(s1.view(np.uint32), s2[: len(s1)].view(np.uint32)) and len(r1) == min(limit, len(r2))
        bad += not ok
        if limit == 1500:
            print("query %2d (%d terms): %d hits, sampled %.3f ms, exact %.3f ms, equal %s" % (qi, len(toks), len(r2), t1 * 1e3, t2 * 1e3, ok), flush=True)
print("mismatches %d, %s" % (bad, g.segments()))
assert bad == 0
