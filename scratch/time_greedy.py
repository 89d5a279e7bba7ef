"""The greedy-MMR kernel alone at one pool size: P = 300 candidates, k picks (run under rocprofv3 --kernel-trace --stats;
k = 2 prices everything in front of the chain, k = 100 the chain)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rlr = importlib.import_module("rust-local-rag_amd")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 300
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
DIM = int(sys.argv[3]) if len(sys.argv) > 3 else 768
ix = rlr.GpuIndex(DIM)
ix.fill_synthetic(20_000, seed=0x5EED0002, n_clusters=40)
rng = np.random.default_rng(1)
for i in range(200):
    pool = rng.permutation(20_000)[:P].astype(np.uint64)
    sc = np.sort(rng.random(P).astype(np.float32))[::-1].copy()
    ix.mmr_select(pool, sc, k, 0.3)
