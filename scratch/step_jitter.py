import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
rlr = importlib.import_module("rust-local-rag_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
prof = len(sys.argv) > 2 and sys.argv[2] == "prof"
ix = rlr.GpuIndex(768, "f32", device=0)
ix.fill_synthetic(n, 3)
rng = np.random.default_rng(0)
qs = np.stack([rlr.normalize(rng.standard_normal(768).astype(np.float32)) for _ in range(64)])
for i in range(10): ix.search_topk(qs[i], 100)
ix.profile_enable(prof)
ts = []
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
t_start = time.perf_counter()
for i in range(steps):
    t0 = time.perf_counter(); ix.search_topk(qs[i % 64], 100); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
med = np.median(ts)
print("n", n, "prof", prof, "median %.1f mean %.1f min %.1f max %.1f" % (med, ts.mean(), ts.min(), ts.max()))
out = [(i, round(float(t), 1)) for i, t in enumerate(ts) if t > 1.15 * med]
print("outliers (>1.15 median):", len(out), out[:40])
cum = np.cumsum(ts) / 1e3
print("outlier start times ms:", [round(float(cum[i] - ts[i] / 1e3), 1) for i, _ in out[:40]])
