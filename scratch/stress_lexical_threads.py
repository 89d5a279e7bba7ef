"""many caller threads (more than workspaces) hammering LexicalIndex.score / RagEngine text search; looks for crashes and
wrong answers.  python scratch/stress_lexical_threads.py <rounds>"""
import importlib, sys, threading, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
rlr = importlib.import_module("rust-local-rag_amd")
lex = importlib.import_module("rust-local-rag_amd.lexical")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = np.random.default_rng(0)
V = 5000
vocab = np.array([f"t{i:05d}" for i in range(V)])
zipf = 1.0 / np.arange(1, V + 1); zipf /= zipf.sum()
n, dim = 60000, 64
eng = rlr.RagEngine(dim)
for b0 in range(0, n, 10000):
    words = rng.choice(V, size=(10000, 20), p=zipf)
    eng.add_document(f"d{b0}", [" ".join(vocab[w]) for w in words], rng.standard_normal((10000, dim)).astype(np.float32))
queries = [" ".join(vocab[rng.choice(V, size=int(rng.integers(1, 6)), p=zipf)]) for _ in range(40)]
embs = rng.standard_normal((40, dim)).astype(np.float32)
want_lex = {q: eng.lexical.score(q, 500) for q in queries}
want_srch = {i: [(r.row, r.score) for r in eng.search_with_diversity(embs[i], 20, 0.3, query_text=queries[i])] for i in range(40)}
errors = []
def worker(tid, iters):
    try:
        for it in range(iters):
            i = (tid * 7 + it) % 40
            if (tid + it) % 2:
                r, s = eng.lexical.score(queries[i], 500)
                wr, ws = want_lex[queries[i]]
                if not (np.array_equal(r, wr) and np.array_equal(s.view(np.uint32), ws.view(np.uint32))):
                    errors.append(("lex", tid, it))
            else:
                got = [(r.row, r.score) for r in eng.search_with_diversity(embs[i], 20, 0.3, query_text=queries[i])]
                if got != want_srch[i]:
                    errors.append(("search", tid, it))
    except Exception as e:
        errors.append((tid, repr(e)))
for rd in range(rounds):
    for nt in (4, 12, 16, 32):
        ts = [threading.Thread(target=worker, args=(t, int(sys.argv[2]) if len(sys.argv) > 2 else 60)) for t in range(nt)]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        print("round %d, %2d threads: %.2f s, errors %d" % (rd, nt, time.perf_counter() - t0, len(errors)), flush=True)
    assert not errors, errors[:5]
print("stress ok", eng.lexical.segments())
