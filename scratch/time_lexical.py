import importlib, sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
lex = importlib.import_module("rust-local-rag_amd.lexical")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(0)
V = 50000
vocab = np.array([f"t{i:05d}" for i in range(V)])
zipf = 1.0 / np.arange(1, V + 1); zipf /= zipf.sum()
g = lex.LexicalIndex(0)
t0 = time.perf_counter()
B = 10000
for b0 in range(0, n, B):
    words = rng.choice(V, size=(B, 30), p=zipf)
    for i in range(B):
        g.add_tokens(b0 + i, vocab[words[i]])
print("built %d docs in %.1f s" % (n, time.perf_counter() - t0), g.info(), flush=True)
t0 = time.perf_counter(); g.score_tokens(["t00000"], 10); print("first score (commit + upload) %.2f s" % (time.perf_counter() - t0))
for name, toks in [("rare 1 term", ["t20000"]), ("mid 3 terms", ["t00100", "t00200", "t00300"]),
                   ("common 1 term", ["t00001"]), ("common+mid 5 terms", ["t00001", "t00002", "t00010", "t00100", "t01000"])]:
    for lim in (500, 1500):
        g.score_tokens(toks, lim)
        t0 = time.perf_counter()
        for _ in range(20):
            r, s = g.score_tokens(toks, lim)
        dt = (time.perf_counter() - t0) / 20
        print("%-20s limit %4d: %7.1f us, %d results, top %.4f" % (name, lim, dt * 1e6, len(r), s[0] if len(s) else 0))

# concurrent callers: each call has its own workspace and stream (csrc/lexical.hip), so the waits overlap
import threading
mix = [["t20000"], ["t00100", "t00200", "t00300"], ["t00001"], ["t00001", "t00002", "t00010", "t00100", "t01000"]]
def run(n_threads, per=200):
    def w(tid):
        for i in range(per):
            g.score_tokens(mix[(tid + i) % 4], 500)
    ts = [threading.Thread(target=w, args=(t,)) for t in range(n_threads)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    print("%2d caller threads: %8.0f score calls/s" % (n_threads, n_threads * per / dt), flush=True)
# --serial: one caller only.  Used for profile collection: several host threads submitting under rocprofv3's queue
# interception trip a ROCr 7.2 / rocprofiler-sdk defect (profiles/r03_profiled_stress_aborts.md); unprofiled, the full sweep runs.
for nt in ((1,) if "--serial" in sys.argv else (1, 2, 4, 8, 16, 32)):
    run(nt)

# ingest loop: one appended chunk, then a search (the commit rebuilds only the appended segment)
ts = []
for i in range(50):
    g.add_tokens(n + i, vocab[rng.choice(V, size=30, p=zipf)])
    t0 = time.perf_counter(); g.score_tokens(["t00100", "t00200"], 100); ts.append(time.perf_counter() - t0)
print("search right after appending one chunk: median %.2f ms, max %.2f ms   segments %s" % (np.median(ts) * 1e3, max(ts) * 1e3, g.segments()), flush=True)
g.add_tokens(5, vocab[rng.choice(V, size=30, p=zipf)])
t0 = time.perf_counter(); g.score_tokens(["t00100", "t00200"], 100); print("search after replacing an old chunk (full rebuild): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
# removal (a document re-ingest drops its old chunks first): both posting segments are compacted on the device
for rows_ in ([7], list(range(1000, 1100)), list(range(50000, 50000 + 5000, 5))):
    t0 = time.perf_counter(); g.remove_rows(rows_); t_rm = time.perf_counter() - t0
    t0 = time.perf_counter(); g.score_tokens(["t00100", "t00200"], 100); t_sc = time.perf_counter() - t0
    print("remove %4d rows: %.2f ms, search right after: %.2f ms   %s" % (len(rows_), t_rm * 1e3, t_sc * 1e3, g.segments()), flush=True)
