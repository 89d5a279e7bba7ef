"""One GPU's share of the headline at 8 GPUs (1.25 M x 768 f32, top-100) through the sharded code path at world 1, 300 steps:
wall time per step, for `rocprofv3 --kernel-trace` (scratch/prof_shard.sh prints the median step's kernel timeline).
RLR_SHARD_ROWS / RLR_SHARD_MODE=plain|sharded"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
rlr = importlib.import_module("rust-local-rag_amd")
sharded = importlib.import_module("rust-local-rag_amd.sharded")
n, dim, k = int(os.environ.get("RLR_SHARD_ROWS", "1250000")), 768, 100
mode = os.environ.get("RLR_SHARD_MODE", "sharded")
sh = sharded.ShardedIndex(dim, n, "f32", device=0, rank=0, world=1)
sh.fill_synthetic(0x5EED0003)
rng = np.random.default_rng(0)
qs = np.stack([rlr.normalize(rng.standard_normal(dim).astype(np.float32)) for _ in range(340)])
import gc
gc.collect(); gc.freeze()
step = (lambda q: sh.search_topk(q, k)) if mode == "sharded" else (lambda q: sh.index.search_topk(q, k))
for i in range(40):
    step(qs[i])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(300):
    step(qs[40 + i])
torch.cuda.synchronize()
print("%s: %d rows, %.1f us per step" % (mode, n, (time.perf_counter() - t0) / 300 * 1e6))
