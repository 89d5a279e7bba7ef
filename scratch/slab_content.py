"""Is a slab's scan rate a property of its memory, or of time / content?  ONE 10 M x 768 index on selected 1 GiB slabs; per-slab
scan rate right after the fill, after a second fill with another seed, after a third with the first seed again."""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RLR_ROWS_ALLOC_NOW"] = os.environ.get("RLR_SLAB_POLICY", "select")
os.environ["RLR_ROWS_ALLOC_LOG"] = "1"
rlr = importlib.import_module("rust-local-rag_amd")
n, dim = 10_000_000, 768
ix = rlr.GpuIndex(dim)
def slabs():
    out = []
    for sl in range(28):
        os.environ["RLR_PROBE_OFF_MIB"] = str(sl * 1024); os.environ["RLR_PROBE_LEN_MIB"] = "1024"
        out.append(round(ix.probe_bandwidth(2, 20)[0] / 8000, 3))
    os.environ.pop("RLR_PROBE_OFF_MIB"); os.environ.pop("RLR_PROBE_LEN_MIB")
    return out
for tag, seed, ncl in (("seed A", 1, 0), ("seed B", 2, 0), ("seed A again", 1, 0), ("one tight cluster", 3, 1 | 0x80000000), ("seed A third", 1, 0)):
    ix.fill_synthetic(n, seed=seed, n_clusters=ncl)
    s1 = slabs()
    whole = round(ix.probe_bandwidth(2, 10)[0] / 8000, 4)
    time.sleep(1.0)
    s2 = slabs()
    print(json.dumps({"content": tag, "whole": whole, "slabs": s1, "slabs_1s_later": s2}), flush=True)
