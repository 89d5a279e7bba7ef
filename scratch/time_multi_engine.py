"""The engine-level entry points over rlr_multi (rlr_multi_engine_*) on the one-GPU box: world = 1 (RCCL exchange) and four
shards on the one device (host merge; the winner rows of the MMR still travel device to device), against the single-index engine.
  * one query, search_with_diversity(top-100, lambda 0.3) over a 1.25 M x 768 f32 shard (10 M / 8: BASELINE config 4's per-GPU
    share at the headline size);
  * a 256-query batch with MMR 0.7 over 1 M x 1024 binary16 rows (config 5's shape, reduced rows so the script runs in seconds)."""
import importlib, sys, time, json, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rlr = importlib.import_module("rust-local-rag_amd")
rng = np.random.default_rng(3)
out = {}

def bench(f, reps=200, warm=20):
    for i in range(warm): f(i)
    t0 = time.perf_counter()
    for i in range(reps): f(i)
    return (time.perf_counter() - t0) / reps * 1e3

# ---- one query, diversity, f32
n, dim, k = 1_250_000, 768, 100
qs = rng.standard_normal((64, dim)).astype(np.float32)
eng = rlr.RagEngine(dim); eng.index.fill_synthetic(n, seed=0x5EED0003); eng._chunks = [None] * n
import ctypes as C
N = rlr._native
def single(i):
    q = qs[i % 64]
    hits = (N.SearchHitC * 300)(); cnt = C.c_uint32()
    N.check(N.lib().rlr_engine_search_with_diversity(eng.index.handle, q.ctypes.data_as(N.f32p), dim, k, 0.3, None, None, None, 0, hits, 300, C.byref(cnt)))
    return cnt.value
out["one_query_diversity"] = {"rows": n, "single_index_fused_ms": bench(single)}
for name, devs, mode in (("multi_world1_rccl", [0], "rccl"), ("multi_world1_host_merge", [0], "host"), ("multi_4_shards_one_device_host_merge", [0, 0, 0, 0], "host")):
    mi = rlr.MultiGpuIndex(dim, devs); mi.fill_synthetic(n, seed=0x5EED0003); mi.set_exchange(mode)
    a = mi.engine_search_with_diversity(qs[0], k, 0.3)
    ref = eng.search_with_diversity(qs[0], k, 0.3) if False else None
    mi.stats(reset=True)
    ms = bench(lambda i: mi.engine_search_with_diversity(qs[i % 64], k, 0.3))
    st = mi.stats()
    out["one_query_diversity"][name + "_ms"] = ms
    out["one_query_diversity"][name + "_mmr_exchange_ms"] = st["mmr_exchange_ms"] / max(st["n_mmr_exchanges"], 1)
    out["one_query_diversity"][name + "_mmr_exchange_bytes"] = st["mmr_exchange_bytes"] // max(st["n_mmr_exchanges"], 1)
    mi.close()
eng.close()

# ---- a batch with MMR, binary16
n, dim, nq, k, lam = 1_000_000, 1024, 256, 100, 0.7
qb = rng.standard_normal((nq, dim)).astype(np.float32)
eng = rlr.RagEngine(dim, "f16"); eng.index.fill_synthetic(n, seed=0x5EED0005, n_clusters=500); eng._chunks = [None] * n
eng.index.enable_batch_image(True)
def single_batch(i):
    cap = 300
    hits = (N.SearchHitC * (cap * nq))(); cnt = np.zeros(nq, np.uint32)
    N.check(N.lib().rlr_engine_search_with_diversity_batch(eng.index.handle, qb.ctypes.data_as(N.f32p), dim, nq, k, lam, None, hits, cap, cnt.ctypes.data_as(N.u32p)))
    return hits, cnt
h1, c1 = single_batch(0)
rows1 = np.frombuffer(h1, dtype=rlr.MultiGpuIndex._HIT, count=300 * nq).reshape(nq, 300)["row"].copy()
out["batch_diversity"] = {"rows": n, "queries": nq, "single_index_ms": bench(single_batch, reps=5, warm=2)}
eng.close()
for name, devs, mode in (("multi_world1_rccl", [0], "rccl"), ("multi_4_shards_one_device_host_merge", [0, 0, 0, 0], "host")):
    mi = rlr.MultiGpuIndex(dim, devs, "f16"); mi.fill_synthetic(n, seed=0x5EED0005, n_clusters=500); mi.enable_batch_image(1); mi.set_exchange(mode)
    got = mi.engine_search_with_diversity_batch(qb, k, lam)
    same = all(np.array_equal(got[q]["row"], rows1[q, : len(got[q])]) for q in range(nq))
    mi.stats(reset=True)
    ms = bench(lambda i: mi.engine_search_with_diversity_batch(qb, k, lam), reps=5, warm=2)
    st = mi.stats()
    out["batch_diversity"][name + "_ms"] = ms
    out["batch_diversity"][name + "_same_hits_as_single_index"] = bool(same)
    out["batch_diversity"][name + "_mmr_exchange_ms"] = st["mmr_exchange_ms"] / max(st["n_mmr_exchanges"], 1)
    out["batch_diversity"][name + "_mmr_exchange_bytes"] = st["mmr_exchange_bytes"] // max(st["n_mmr_exchanges"], 1)
    mi.close()
print(json.dumps(out, indent=1))
