"""batched MMR alone at BASELINE config 5's pool shape: 1024 pools of 300 x 1024-d binary16 rows, top-100, lambda 0.7 (the pools come
from a batched search over a 1 M-row synthetic corpus).   python scratch/time_mmr_f16.py [reps]"""
import importlib, sys, time, json
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
rlr = importlib.import_module("rust-local-rag_amd")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n, dim, nq, P, k, lam = 1_000_000, 1024, 1024, 300, 100, 0.7
ix = rlr.GpuIndex(dim, "f16")
ix.fill_synthetic(n, seed=0x5EED0005, n_clusters=500)
rng = np.random.default_rng(5)
qn = np.stack([rlr.normalize(q) for q in rng.standard_normal((nq, dim)).astype(np.float32)])
r, c = ix.search_topk(qn, P)
sc = (np.float32(0.7) * c).astype(np.float32)
sizes = np.full(nq, P, np.uint32)
ix.mmr_select_batch(r, sc, sizes, k, lam)
ix.profile_read(reset=True); ix.profile_enable(True)
t0 = time.perf_counter()
for _ in range(reps):
    order, mmr, nn = ix.mmr_select_batch(r, sc, sizes, k, lam)
dt = (time.perf_counter() - t0) / reps
p = ix.profile_read()
print(json.dumps({"pools": nq, "P": P, "dim": dim, "call_ms": dt * 1e3, "mmr_kernels_ms": p.mmr_ms / reps, "picks": int(nn.sum())}))
