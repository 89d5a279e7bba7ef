"""The BM25 chain of bench.py's text leg alone (no scan beside it): 100 k chunks of 40 Zipf words, 6-word queries, limit 1500.
Under rocprofv3 --kernel-trace --stats this gives the uncontended kernel durations."""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lex_mod = importlib.import_module("rust-local-rag_amd.lexical")
n = 100_000
rng = np.random.default_rng(0x5EED0012)
V = 20000
vocab = np.array([f"t{i:05d}" for i in range(V)])
zipf = 1.0 / np.arange(1, V + 1); zipf /= zipf.sum()
lx = lex_mod.LexicalIndex(0)
for b0 in range(0, n, 10000):
    words = rng.choice(V, size=(10000, 40), p=zipf)
    for i in range(10000):
        lx.add_tokens(b0 + i, vocab[words[i]])
toks = [list(vocab[rng.choice(V, size=6, p=zipf)]) for _ in range(220)]
for i in range(20):
    lx.score_tokens(toks[i], 1500)
t0 = time.perf_counter()
for i in range(20, 220):
    lx.score_tokens(toks[i], 1500)
print("score_tokens alone: %.1f us per call" % ((time.perf_counter() - t0) / 200 * 1e6))
lx.close()
