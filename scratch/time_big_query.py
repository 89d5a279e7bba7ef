"""the large-candidate path (guard band > 4096 entries): dense cluster around the query, 8-bit nomination.
RLR_BIG_QUERY_SORT=1 selects the old finish (one-lane re-score + global bitonic sort)."""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rlr = importlib.import_module("rust-local-rag_amd")
n, dim = 1_000_000, 768
ix = rlr.GpuIndex(dim)
ix.fill_synthetic(n, seed=5, n_clusters=0)
rng = np.random.default_rng(1)
q = rlr.normalize(rng.standard_normal(dim).astype(np.float32))
tight = q[None, :] + np.float32(0.02) * rng.standard_normal((40000, dim)).astype(np.float32)
tight /= np.linalg.norm(tight, axis=1, keepdims=True).astype(np.float32)
ix.append(tight.astype(np.float32))
ix.enable_batch_image(False, q8=True)
for k in (100, 1000, 3000):
    ix.search_topk(q, k)
    ix.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(10): r, c = ix.search_topk(q, k)
    dt = (time.perf_counter() - t0) / 10
    p = ix.profile_read()
    print("k %4d: %.3f ms per query, retries %d of %d, candidates %.0f" % (k, dt * 1e3, p.n_retries, p.n_searches, p.n_candidates / max(p.n_searches, 1)), flush=True)
