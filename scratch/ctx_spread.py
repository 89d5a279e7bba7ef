"""Rows or scratch?  ONE 10 M x 768 index, four search contexts (each with its own score array / histograms / query buffer),
sequential calls cycling through them (RLR_CTX_ROTATE): a period-4 pattern in the scan time means the placement of the per-context
buffers, not of the rows, decides where in the 0.86-0.89 band a scan lands."""
import importlib, json, os, sys, threading
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RLR_CTX_ROTATE"] = "1"
rlr = importlib.import_module("rust-local-rag_amd")
n, dim = int(os.environ.get("RLR_SPREAD_ROWS", "10000000")), 768
rng = np.random.default_rng(3)
qs = rng.standard_normal((64, dim)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
for trial in range(3):
    ix = rlr.GpuIndex(dim)
    ix.fill_synthetic(n, seed=0x5EED0003)
    ths = [threading.Thread(target=lambda i=i: [ix.search_topk(qs[i], 100) for _ in range(3)]) for i in range(4)]
    [t.start() for t in ths]; [t.join() for t in ths]
    ix.profile_enable(True)
    out = []
    for i in range(48):
        ix.profile_read(reset=True)
        ix.search_topk(qs[i], 100)
        p = ix.profile_read()
        out.append(round(n * dim * 4 / (p.scan_ms * 1e-3) / 8e12, 4))
    by_ctx = [round(float(np.mean(out[j::4])), 4) for j in range(4)]
    print(json.dumps({"trial": trial, "frac_by_context_slot": by_ctx, "spread_within_slot": [round(float(np.ptp(out[j::4])), 4) for j in range(4)]}), flush=True)
    ix.close()
