"""One GPU's share of BASELINE config 5: 6.25M x 1024-d fp16 rows (50M / 8), 1024 batched queries,
top_k=100, MMR lambda=0.7 -- batched search (pool 300) + batched MMR through the engine API."""
import importlib, sys, time, json
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
rlr = importlib.import_module("rust-local-rag_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6_250_000
dim, nq, k, lam = 1024, 1024, 100, 0.7
eng = rlr.RagEngine(dim, "f16")
t0 = time.perf_counter(); eng.index.fill_synthetic(n, seed=0x5EED0005, n_clusters=500); fill = time.perf_counter() - t0
eng._chunks = [None] * n
rng = np.random.default_rng(5)
qs = rng.standard_normal((nq, dim)).astype(np.float32)
qn = np.stack([rlr.normalize(q) for q in qs])
ix = eng.index
if "--image" in sys.argv:
    ix.enable_batch_image(True)
# raw batched top-308 search
ix.search_topk(qn[:64], 308)
if "--cold" not in sys.argv:  # steady state: the first full-size call grows the per-call workspaces (hipMalloc / hipHostMalloc)
    ix.search_topk(qn, 308)
ix.profile_read(reset=True); ix.profile_enable(True)
t0 = time.perf_counter(); r, c = ix.search_topk(qn, 308); t_search = time.perf_counter() - t0
p = ix.profile_read(); ix.profile_enable(False)
# batched MMR alone
sc = (np.float32(0.7) * c).astype(np.float32)
sizes = np.full(nq, 308, np.uint32)
if "--cold" not in sys.argv:
    ix.mmr_select_batch(r[:, :300].copy(), sc[:, :300].copy(), np.full(nq, 300, np.uint32), k, lam)
rr, ss, zz = r[:, :300].copy(), sc[:, :300].copy(), np.full(nq, 300, np.uint32)
ix.profile_read(reset=True); ix.profile_enable(True)
t0 = time.perf_counter(); order, mmr, nn = ix.mmr_select_batch(rr, ss, zz, k, lam); t_mmr = time.perf_counter() - t0
pm = ix.profile_read(); ix.profile_enable(False)
mmr_info = {"mmr_kernels_ms": pm.mmr_ms}
# single-path spot check of 3 queries
ok = True
for i in (0, 511, 1023):
    r1, c1 = ix.search_topk(qn[i], 308)
    ok &= bool(np.array_equal(r1[0], r[i]) and np.array_equal(c1[0].view(np.uint32), c[i].view(np.uint32)))
print(json.dumps({"rows": n, "dim": dim, "dtype": "f16", "queries": nq, "fill_s": round(fill, 2),
                  "batched_search_top308_ms": t_search * 1e3, "qps_search": nq / t_search,
                  "gemm_ms": p.batch_gemm_ms, "other_ms": p.batch_other_ms, "fallbacks": p.n_batch_fallbacks,
                  "gemm_GBps": p.batch_gemm_bytes / (p.batch_gemm_ms * 1e-3) / 1e9 if p.batch_gemm_ms else None,
                  "gemm_TFLOPs": p.batch_gemm_flops / (p.batch_gemm_ms * 1e-3) / 1e12 if p.batch_gemm_ms else None,
                  "mmr_batch_ms": t_mmr * 1e3, "mmr": mmr_info, "batched_equals_single_path": ok}))
