"""Does the way the row matrix is allocated decide where in the 0.86-0.89 band the headline scan lands?  10 M x 768 f32 indexes
created under different RLR_ROWS_ALLOC_NOW policies in ONE process (several alive at a time), each timed: the scan stage (HIP
events, 30 queries) and the read-only probe.  Prints one JSON line per index."""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RLR_ROWS_ALLOC_LOG"] = "1"
rlr = importlib.import_module("rust-local-rag_amd")
n, dim = int(os.environ.get("RLR_SPREAD_ROWS", "10000000")), 768
rng = np.random.default_rng(3)
qs = rng.standard_normal((35, dim)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
rounds = [["plain", "plain", "plain", "plain", "align:2", "align:1024"],
          ["round:1024", "vmm:2", "vmm:256", "vmm:1024", "vmm:4096", "plain"],
          ["vmm:1024", "vmm:1024", "align:1024", "align:1024", "vmm:32768", "plain"]]
if os.environ.get("RLR_SPREAD_ROUNDS"):
    rounds = [r.split(",") for r in os.environ["RLR_SPREAD_ROUNDS"].split(";")]
for rnd, pols in enumerate(rounds):
    ixs = []
    for pol in pols:
        os.environ["RLR_ROWS_ALLOC_NOW"] = pol
        ix = rlr.GpuIndex(dim)
        try:
            ix.fill_synthetic(n, seed=0x5EED0003)
        except Exception as e:
            print(json.dumps({"round": rnd, "policy": pol, "error": str(e)}), flush=True)
            ix.close()
            continue
        ixs.append((pol, ix))
    for rep in range(2):
        for pol, ix in ixs:
            for q in qs[:5]:
                ix.search_topk(q, 100)
            ix.profile_read(reset=True); ix.profile_enable(True)
            for q in qs[5:]:
                ix.search_topk(q, 100)
            ix.profile_enable(False)
            p = ix.profile_read()
            rd = ix.probe_bandwidth(0, 3)[0] if rep == 1 else None
            print(json.dumps({"round": rnd, "rep": rep, "policy": pol, "scan_ms": round(p.scan_ms / max(p.n_scan_launches, 1), 4),
                              "frac": round(n * dim * 4 / (p.scan_ms / max(p.n_scan_launches, 1) * 1e-3) / 8e12, 4),
                              "probe_read_GBps": rd and round(rd, 1)}), flush=True)
    for _, ix in ixs:
        ix.close()
