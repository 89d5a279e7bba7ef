"""Does the scan rate drift while the device warms up?  One 10 M x 768 f32 index, back-to-back searches for 20 s, the scan stage's
fraction of the HBM peak (HIP events) averaged per second.  RLR_ROWS_ALLOC_NOW picks the allocation policy."""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rlr = importlib.import_module("rust-local-rag_amd")
n, dim = 10_000_000, 768
rng = np.random.default_rng(3)
qs = rng.standard_normal((64, dim)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
ix = rlr.GpuIndex(dim)
ix.fill_synthetic(n, seed=0x5EED0003)
ix.profile_enable(True)
t0 = time.time(); out = []; i = 0
for sec in range(int(os.environ.get("RLR_DRIFT_S", "20"))):
    ix.profile_read(reset=True)
    while time.time() - t0 < sec + 1:
        ix.search_topk(qs[i % 64], 100); i += 1
    p = ix.profile_read()
    out.append(round(n * dim * 4 / (p.scan_ms / p.n_scan_launches * 1e-3) / 8e12, 4))
print(json.dumps({"policy": os.environ.get("RLR_ROWS_ALLOC_NOW", "default"), "frac_per_second": out}))
