"""Many caller threads (more than cores per GPU share, more than search contexts) on ONE index through the polling wait: plain top-k,
search_with_diversity and a sharded step, every answer compared with the single-threaded one.  python scratch/stress_search_threads.py <threads> <iters>"""
import importlib, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rlr = importlib.import_module("rust-local-rag_amd")
n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n, dim = 300_000, 768
eng = rlr.RagEngine(dim)
eng.index.fill_synthetic(n, seed=77, n_clusters=50)
eng._chunks = [rlr.DocumentChunk(str(i), "s", "", i) for i in range(n)]
rng = np.random.default_rng(1)
qs = rng.standard_normal((64, dim)).astype(np.float32)
qn = np.stack([rlr.normalize(q) for q in qs])
want_topk = [eng.index.search_topk(qn[i], 100) for i in range(64)]
want_div = [[(r.row, r.score) for r in eng.search_with_diversity(qs[i], 20, 0.3)] for i in range(64)]
errors = []
def worker(tid):
    try:
        for it in range(iters):
            i = (tid * 7 + it) % 64
            if it % 2:
                r, c = eng.index.search_topk(qn[i], 100)
                if not (np.array_equal(r, want_topk[i][0]) and np.array_equal(c.view(np.uint32), want_topk[i][1].view(np.uint32))):
                    errors.append(("topk", tid, it, i))
            else:
                got = [(r.row, r.score) for r in eng.search_with_diversity(qs[i], 20, 0.3)]
                if got != want_div[i]:
                    errors.append(("div", tid, it, i))
    except Exception as e:
        errors.append(("exc", tid, repr(e)))
t0 = time.time()
ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
[t.start() for t in ths]; [t.join() for t in ths]
el = time.time() - t0
print("threads %d iters %d: %d errors %s; %.0f calls/s" % (n_threads, iters, len(errors), errors[:3], n_threads * iters / el))
eng.close()
