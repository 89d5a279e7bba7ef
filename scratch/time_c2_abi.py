"""C2 through the C ABI only (no Python result objects): rlr_engine_search_with_diversity timed per call,
plus the profile's kernel sums.  100k x 768 f32, top_k=100, lambda=0.3."""
import importlib, sys, time, json, ctypes as C
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
rlr = importlib.import_module("rust-local-rag_amd")
N = rlr._native
n, dim, k, lam = (int(sys.argv[2]) if len(sys.argv) > 2 else 100_000), 768, (int(sys.argv[1]) if len(sys.argv) > 1 else 100), 0.3
ix = rlr.GpuIndex(dim)
ix.fill_synthetic(n, seed=0x5EED0002, n_clusters=200)
rng = np.random.default_rng(1)
qs = rng.standard_normal((300, dim)).astype(np.float32)
cap = 300
hits = (N.SearchHitC * cap)()
nn = C.c_uint32()
L = N.lib()
def call(q):
    N.check(L.rlr_engine_search_with_diversity(ix.handle, q.ctypes.data_as(N.f32p), dim, k, lam, None, None, None, 0, hits, cap, C.byref(nn)))
for i in range(20): call(qs[i])
ix.profile_read(reset=True); ix.profile_enable(True)
t0 = time.perf_counter()
for i in range(20, 270): call(qs[i])
t = (time.perf_counter() - t0) / 250
ix.profile_enable(False)
p = ix.profile_read()
ns = max(p.n_scan_launches, 1)
print(json.dumps({"abi_call_ms_profiled": t * 1e3, "scan": p.scan_ms / ns, "select": p.select_ms / ns, "rescore_sort": p.rescore_ms / ns,
                  "mmr_block": p.mmr_ms / max(p.n_mmr, 1), "n": nn.value}))
t0 = time.perf_counter()
for i in range(20, 270): call(qs[i])
t = (time.perf_counter() - t0) / 250
print(json.dumps({"abi_call_ms": t * 1e3}))
