import importlib, sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
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
import gc
if "gcfreeze" in sys.argv:      # full collection now, then move every surviving object out of the collector's sight
    gc.collect(); gc.freeze()
if "gcoff" in sys.argv:
    gc.disable()
gc_events = []
gc.callbacks.append(lambda phase, info: gc_events.append((phase, info["generation"], time.perf_counter())))
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
full = [(round((t - t_start) * 1e3, 1)) for ph, g, t in gc_events if ph == "start" and g == 2]
print("gen-2 collections started at ms:", full)
print("outlier start times ms:", [round(float(cum[i] - ts[i] / 1e3), 1) for i, _ in out[:40]])
