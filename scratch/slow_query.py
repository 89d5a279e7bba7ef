import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
rlr = importlib.import_module("rust-local-rag_amd")
n = 10_000_000
ix = rlr.GpuIndex(768, "f32", device=0)
ix.fill_synthetic(n, 3)
rng = np.random.default_rng(0)
qs = np.stack([rlr.normalize(rng.standard_normal(768).astype(np.float32)) for _ in range(320)])
for i in range(5): ix.search_topk(qs[i], 100)
ix.profile_enable(True)
for i in [43, 44, 45, 140, 141, 142, 141, 44]:
    ix.profile_read(reset=True)
    t0 = time.perf_counter(); r = ix.search_topk(qs[i], 100); dt = time.perf_counter() - t0
    p = ix.profile_read()
    print(i, "%.1f us" % (dt * 1e6), p)
