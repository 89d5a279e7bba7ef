"""Slow slabs under the scan's own access pattern?  Four 10 M x 768 f32 indexes on 1 GiB physical slabs: the scan kernel over the
whole index and over each 1 GiB slab alone (rlr_index_probe_bandwidth mode 2), fraction of 8 TB/s."""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RLR_ROWS_ALLOC_NOW"] = os.environ.get("RLR_SLAB_POLICY", "vmm:1024")
rlr = importlib.import_module("rust-local-rag_amd")
n, dim = 10_000_000, 768
ixs = []
for i in range(4):
    ix = rlr.GpuIndex(dim)
    ix.fill_synthetic(n, seed=0x5EED0003)
    ixs.append(ix)
for i, ix in enumerate(ixs):
    os.environ.pop("RLR_PROBE_OFF_MIB", None); os.environ.pop("RLR_PROBE_LEN_MIB", None)
    whole = ix.probe_bandwidth(2, 10)[0] / 8000
    slabs = []
    for sl in range(28):
        os.environ["RLR_PROBE_OFF_MIB"] = str(sl * 1024); os.environ["RLR_PROBE_LEN_MIB"] = "1024"
        slabs.append(round(ix.probe_bandwidth(2, 20)[0] / 8000, 3))
    pairs = []
    for sl in range(0, 28, 4):
        os.environ["RLR_PROBE_OFF_MIB"] = str(sl * 1024); os.environ["RLR_PROBE_LEN_MIB"] = "4096"
        pairs.append(round(ix.probe_bandwidth(2, 10)[0] / 8000, 3))
    print(json.dumps({"index": i, "scan_whole": round(whole, 4), "slab_min": min(slabs), "slab_max": max(slabs), "slabs_1GiB": slabs,
                      "runs_4GiB": pairs}), flush=True)
