"""Which part of the scan is placement-sensitive?  Six 10 M x 768 f32 indexes in one process: the real scan (events), the read
probe in the scan's shape, the scan kernel without and with its histogram flush into fresh scratch buffers."""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RLR_PROBE_SHAPE"] = "0"
rlr = importlib.import_module("rust-local-rag_amd")
n, dim = 10_000_000, 768
rng = np.random.default_rng(3)
qs = rng.standard_normal((25, dim)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
ixs = []
for i in range(6):
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
    r = {m: round(ix.probe_bandwidth(m, 10)[0] / 8000, 4) for m in (0, 2, 3, 2, 3)}
    r2 = [round(ix.probe_bandwidth(m, 10)[0] / 8000, 4) for m in (2, 3, 2, 3)]
    print(json.dumps({"index": i, "scan_frac": round(frac, 4), "read_probe": r[0], "scan_nohist,hist x2": r2}), flush=True)
