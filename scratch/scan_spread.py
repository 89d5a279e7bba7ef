"""Where does the run-to-run spread of the one-wave-per-SIMD scan come from?  Four 10 M x 768 f32 indexes in ONE process (four
allocations), the scan stage timed on each in turn, twice round: a spread BETWEEN the indexes that repeats across the rounds is
placement; a spread between the rounds of one index is something else."""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rlr = importlib.import_module("rust-local-rag_amd")
n, dim = 10_000_000, 768
rng = np.random.default_rng(3)
qs = rng.standard_normal((40, dim)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
ixs = []
for i in range(4):
    ix = rlr.GpuIndex(dim)
    ix.fill_synthetic(n, seed=0x5EED0003)
    ixs.append(ix)
out = []
for rnd in range(2):
    for i, ix in enumerate(ixs):
        for q in qs[:5]:
            ix.search_topk(q, 100)
        ix.profile_read(reset=True); ix.profile_enable(True)
        for q in qs:
            ix.search_topk(q, 100)
        ix.profile_enable(False)
        p = ix.profile_read()
        out.append({"round": rnd, "index": i, "scan_ms": round(p.scan_ms / max(p.n_scan_launches, 1), 4)})
print(json.dumps(out))
