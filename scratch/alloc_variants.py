"""Is a slow placement slow for every launch shape of the scan?  Six 10 M x 768 f32 indexes (plain hipMalloc) in one process,
each timed under several RLR_SCAN_VARIANT shapes (r | workgroups per CU << 8 | rows per group << 16)."""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RLR_SCAN_VARIANT_DYN"] = "1"
rlr = importlib.import_module("rust-local-rag_amd")
n, dim = int(os.environ.get("RLR_SPREAD_ROWS", "10000000")), 768
rng = np.random.default_rng(3)
qs = rng.standard_normal((25, dim)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
variants = {"default": 0, "r4 b1 g4": 4 | 1 << 8 | 4 << 16, "r4 b1 g16": 4 | 1 << 8 | 16 << 16, "r4 b1 g32": 4 | 1 << 8 | 32 << 16,
            "r4 b1 g64": 4 | 1 << 8 | 64 << 16, "r4 b2 g8": 4 | 2 << 8 | 8 << 16, "r8 b4 g16": 8 | 4 << 8 | 16 << 16, "r2 b2 g8": 2 | 2 << 8 | 8 << 16}
ixs = []
for i in range(int(os.environ.get("RLR_SPREAD_N", "6"))):
    ix = rlr.GpuIndex(dim)
    ix.fill_synthetic(n, seed=0x5EED0003)
    ixs.append(ix)
for i, ix in enumerate(ixs):
    row = {}
    for name, v in variants.items():
        os.environ["RLR_SCAN_VARIANT"] = str(v)
        for q in qs[:5]:
            ix.search_topk(q, 100)
        ix.profile_read(reset=True); ix.profile_enable(True)
        for q in qs[5:]:
            ix.search_topk(q, 100)
        ix.profile_enable(False)
        p = ix.profile_read()
        row[name] = round(n * dim * 4 / (p.scan_ms / max(p.n_scan_launches, 1) * 1e-3) / 8e12, 4)
    print(json.dumps({"index": i, **row}), flush=True)
