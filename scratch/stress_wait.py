"""Polling wait vs stream synchronisation: every query through the profiled path (hipStreamSynchronize) and then N times through the
polling path; prints how a mismatching result differs.  RLR_STRESS_ROWS / RLR_STRESS_REPS"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
rlr = importlib.import_module("rust-local-rag_amd")
n, dim, k = int(os.environ.get("RLR_STRESS_ROWS", "300000")), 768, 100
reps = int(os.environ.get("RLR_STRESS_REPS", "20"))
ix = rlr.GpuIndex(dim)
ix.fill_synthetic(n, seed=640)
rng = np.random.default_rng(1)
qs = np.stack([rlr.normalize(rng.standard_normal(dim).astype(np.float32)) for _ in range(256)])
ix.profile_enable(True)
ref = [ix.search_topk(q, k) for q in qs]
ix.profile_enable(False)
bad = 0
for rep in range(reps):
    for i, q in enumerate(qs):
        r, c = ix.search_topk(q, k)
        if not (np.array_equal(r, ref[i][0]) and np.array_equal(c.view(np.uint32), ref[i][1].view(np.uint32))):
            bad += 1
            d = np.flatnonzero((r[0] != ref[i][0][0]) | (c[0].view(np.uint32) != ref[i][1][0].view(np.uint32)))
            prev = ref[i - 1][0][0] if i else ref[-1][0][0]
            print(f"rep {rep} query {i}: {d.size} differing slots {d[:12]}; got rows {r[0][d[:6]]} want {ref[i][0][0][d[:6]]} "
                  f"previous query's rows there {prev[d[:6]]}; shape {r.shape}", flush=True)
print("mismatches", bad, "of", reps * len(qs))
