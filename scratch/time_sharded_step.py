import importlib, sys, time, os, ctypes as C
import numpy as np
sys.path.insert(0, '.')
import torch
rlr = importlib.import_module("rust-local-rag_amd")
sharded = importlib.import_module("rust-local-rag_amd.sharded")
N = importlib.import_module("rust-local-rag_amd._native")
n, dim, k = 1_250_000, 768, 100
sh = sharded.ShardedIndex(dim, n, "f32", device=0, rank=0, world=1)
sh.fill_synthetic(3)
rng = np.random.default_rng(0)
qs = np.stack([rlr.normalize(rng.standard_normal(dim).astype(np.float32)) for _ in range(300)])
ix = sh.index
for i in range(20): sh.search_topk(qs[i], k)
# segment timing
t_dev = t_merge = t_tot = t_plain = 0.0
local = torch.zeros((1, k), dtype=torch.int64, device="cuda")
rows_h = np.zeros((1, k), np.uint64); cos_h = np.zeros((1, k), np.float32); n_h = np.zeros(1, np.uint32)
bases = np.zeros(1, np.uint64)
stream = torch.cuda.current_stream().cuda_stream
for i in range(20, 220):
    q = qs[i]
    t0 = time.perf_counter()
    ix.search_topk_device(q, k, local.data_ptr(), stream)
    t1 = time.perf_counter()
    N.check(N.lib().rlr_merge_topk(0, C.c_void_p(local.data_ptr()), 1, 1, k, bases.ctypes.data_as(N.u64p), rows_h.ctypes.data_as(N.u64p), cos_h.ctypes.data_as(N.f32p), n_h.ctypes.data_as(N.u32p), C.c_void_p(stream)))
    t2 = time.perf_counter()
    t_dev += t1 - t0; t_merge += t2 - t1
for i in range(20, 220):
    t0 = time.perf_counter(); ix.search_topk(qs[i], k); t_plain += time.perf_counter() - t0
for i in range(20, 220):
    t0 = time.perf_counter(); sh.search_topk(qs[i], k); t_tot += time.perf_counter() - t0
print("search_topk_device %.1f us | merge_topk %.1f us | sharded.search_topk %.1f us | plain search_topk %.1f us" % (t_dev/200*1e6, t_merge/200*1e6, t_tot/200*1e6, t_plain/200*1e6))
