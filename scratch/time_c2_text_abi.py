"""config 2 with the query text at the C ABI (bench.py's with_query_text leg alone): rlr_engine_search_text per call."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
rlr = importlib.import_module("rust-local-rag_amd")
n, dim, k, lam, steps = 100_000, 768, 100, 0.3, 200
ix = rlr.GpuIndex(dim)
ix.fill_synthetic(n, seed=0x5EED0002, n_clusters=200)
rng = np.random.default_rng(0x5EED0002)
qs = rng.standard_normal((steps + 20, dim)).astype(np.float32)
out = bench.c2_hybrid_leg(rlr, torch, ix, qs, n, dim, k, lam, steps)
print(json.dumps({"text_ms_per_query": out["ms_per_query"], "with_lexical": out["of_them_with_a_lexical_score"]}))
ix.close()
