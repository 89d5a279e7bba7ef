"""The two forms of the BM25 accumulation kernel against each other at a size where a workgroup of the LDS form owns more
than 256 rows (several chunks per term): the same corpus in two indexes, one created with RLR_LEX_TERMS=global, the same
queries, results compared bit for bit; appended rows (second posting segment) included.
python scratch/lex_forms_check.py <docs>"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lex = importlib.import_module("rust-local-rag_amd.lexical")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600_000
rng = np.random.default_rng(5)
V = 30000
vocab = np.array([f"t{i:05d}" for i in range(V)])
zipf = 1.0 / np.arange(1, V + 1); zipf /= zipf.sum()
os.environ.pop("RLR_LEX_TERMS", None)
a = lex.LexicalIndex(0)
os.environ["RLR_LEX_TERMS"] = "global"
b = lex.LexicalIndex(0)
os.environ.pop("RLR_LEX_TERMS", None)
t0 = time.perf_counter()
def add(lo, hi):
    B = 20000
    for b0 in range(lo, hi, B):
        m = min(B, hi - b0)
        words = rng.choice(V, size=(m, 24), p=zipf)
        lens = rng.integers(8, 25, size=m)
        for i in range(m):
            t = vocab[words[i, : lens[i]]]
            a.add_tokens(b0 + i, t)
            b.add_tokens(b0 + i, t)
add(0, n)
print("built 2 x %d docs in %.1f s" % (n, time.perf_counter() - t0), flush=True)
bad = 0
def compare(tag):
    global bad
    for qi in range(16):
        toks = list(vocab[rng.choice(V, size=int(rng.integers(1, 7)), p=zipf)])
        for limit in (200, 1500, 5000):
            t0 = time.perf_counter(); r1, s1 = a.score_tokens(toks, limit); t1 = time.perf_counter() - t0
            t0 = time.perf_counter(); r2, s2 = b.score_tokens(toks, limit); t2 = time.perf_counter() - t0
            ok = np.array_equal(r1, r2) and np.array_equal(s1.view(np.uint32), s2.view(np.uint32))
            bad += not ok
            if limit == 1500 and qi % 4 == 0:
                print("%s query %2d (%d terms): %d hits, lds %.3f ms, global %.3f ms, equal %s" % (tag, qi, len(toks), len(r1), t1 * 1e3, t2 * 1e3, ok), flush=True)
compare("main")
add(n, n + 30000)          # appended rows: the second posting segment
compare("main+appended")
print("mismatches %d" % bad)
assert bad == 0
