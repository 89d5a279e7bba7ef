"""Per-call time of lexical scoring with the BM25 sums in LDS vs in device memory, over corpus sizes.
python scratch/lex_forms_time.py <docs> [<docs> ...]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lex = importlib.import_module("rust-local-rag_amd.lexical")
V = 30000
vocab = np.array([f"t{i:05d}" for i in range(V)])
zipf = 1.0 / np.arange(1, V + 1); zipf /= zipf.sum()
for n in [int(x) for x in sys.argv[1:]] or [100_000]:
    rng = np.random.default_rng(5)
    os.environ.pop("RLR_LEX_TERMS", None)
    a = lex.LexicalIndex(0)
    os.environ["RLR_LEX_TERMS"] = "global"
    b = lex.LexicalIndex(0)
    os.environ.pop("RLR_LEX_TERMS", None)
    B = 20000
    for b0 in range(0, n, B):
        m = min(B, n - b0)
        words = rng.choice(V, size=(m, 24), p=zipf)
        for i in range(m):
            a.add_tokens(b0 + i, vocab[words[i]]); b.add_tokens(b0 + i, vocab[words[i]])
    qs = [list(vocab[rng.choice(V, size=6, p=zipf)]) for _ in range(40)]
    out = {}
    for name, g in (("lds", a), ("global", b)):
        for q in qs[:5]: g.score_tokens(q, 1500)
        t0 = time.perf_counter()
        for rep in range(5):
            for q in qs: g.score_tokens(q, 1500)
        out[name] = (time.perf_counter() - t0) / 200 * 1e6
    print("n %8d: lds %.1f us  global %.1f us per score_tokens(6 terms, limit 1500)" % (n, out["lds"], out["global"]), flush=True)
    a.close(); b.close()
