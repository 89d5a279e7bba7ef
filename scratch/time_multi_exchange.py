"""rlr_multi at world = 1 on the one-GPU box: host merge vs RCCL all-gather + merge kernel, per single query
(10 M / 8 = 1.25 M-row shard, top-100), and the plain single index for reference."""
import importlib, sys, time, json, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rlr = importlib.import_module("rust-local-rag_amd")
n, dim, k = 1_250_000, 768, 100
rng = np.random.default_rng(3)
qs = np.stack([rlr.normalize(rng.standard_normal(dim).astype(np.float32)) for _ in range(64)])
mi = rlr.MultiGpuIndex(dim, [0]); mi.fill_synthetic(n, seed=0x5EED0003)
ix = rlr.GpuIndex(dim); ix.fill_synthetic(n, seed=0x5EED0003)
def bench(f, reps=300):
    for i in range(30): f(qs[i % 64])
    t0 = time.perf_counter()
    for i in range(reps): f(qs[i % 64])
    return (time.perf_counter() - t0) / reps * 1e3
out = {"rows": n, "single_index_ms": bench(lambda q: ix.search_topk(q, k)),
       "multi_host_merge_ms": bench(lambda q: mi.search_topk(q, k))}
mi.set_exchange("rccl")
out["multi_rccl_allgather_merge_ms"] = bench(lambda q: mi.search_topk(q, k))
a = mi.search_topk(qs[:4], k); mi.set_exchange("host"); b = mi.search_topk(qs[:4], k)
out["same_results"] = bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)))
print(json.dumps(out))
