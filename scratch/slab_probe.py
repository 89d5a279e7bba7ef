"""Slow slabs or a collective effect?  10 M x 768 f32 indexes on 1 GiB physical slabs (RLR_ROWS_ALLOC_NOW=vmm:1024): the scan's
fraction of the HBM peak over the whole index, then the scan-shaped read probe over each 1 GiB slab on its own."""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RLR_ROWS_ALLOC_NOW"] = os.environ.get("RLR_SLAB_POLICY", "vmm:1024")
os.environ["RLR_PROBE_SHAPE"] = "0"
rlr = importlib.import_module("rust-local-rag_amd")
n, dim = 10_000_000, 768
rng = np.random.default_rng(3)
qs = rng.standard_normal((25, dim)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
ixs = []
for i in range(4):
    ix = rlr.GpuIndex(dim)
    ix.fill_synthetic(n, seed=0x5EED0003)
    ixs.append(ix)
for i, ix in enumerate(ixs):
    for q in qs[:5]:
        ix.search_topk(q, 100)
    ix.profile_read(reset=True); ix.profile_enable(True)
    for q in qs[5:]:
        ix.search_topk(q, 100)
    ix.profile_enable(False)
    p = ix.profile_read()
    frac = n * dim * 4 / (p.scan_ms / p.n_scan_launches * 1e-3) / 8e12
    os.environ.pop("RLR_PROBE_OFF_MIB", None); os.environ.pop("RLR_PROBE_LEN_MIB", None)
    whole = ix.probe_bandwidth(0, 3)[0]
    slabs = []
    for sl in range(28):
        os.environ["RLR_PROBE_OFF_MIB"] = str(sl * 1024); os.environ["RLR_PROBE_LEN_MIB"] = "1024"
        slabs.append(round(ix.probe_bandwidth(0, 8)[0]))
    print(json.dumps({"index": i, "scan_frac": round(frac, 4), "probe_shape0_whole": round(whole), "slabs_min": min(slabs), "slabs_max": max(slabs),
                      "slabs": slabs}), flush=True)
