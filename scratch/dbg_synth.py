import sys, importlib, numpy as np
sys.path.insert(0,'.')
rlr = importlib.import_module("rust-local-rag_amd")
from oracle import oracle as O
for dim,dtype,ncl in ((768,"f32",0),(96,"f32",5)):
    ix = rlr.GpuIndex(dim, dtype)
    ix.fill_synthetic(300, seed=42, row0=1000, n_clusters=ncl)
    got = ix.fetch_rows(np.arange(300))
    want = O.synth_rows(300, dim, seed=42, row0=1000, n_clusters=ncl)
    d = got.view(np.uint32).astype(np.int64) - want.view(np.uint32).astype(np.int64)
    bad = np.argwhere(d!=0)
    print(dim, "mismatch elems", len(bad), "rows", len(set(bad[:,0])), "max ulp", np.abs(d).max())
    print(bad[:10], d[d!=0][:10])
# normalize path
rng=np.random.default_rng(5)
raw=(rng.standard_normal((500,768))*3).astype(np.float32)
ix=rlr.GpuIndex(768); ix.upload(raw, normalize=True)
got=ix.fetch_rows(np.arange(500)); want=np.stack([O.normalize(r) for r in raw])
d = got.view(np.uint32).astype(np.int64) - want.view(np.uint32).astype(np.int64)
bad=np.argwhere(d!=0); print("normalize mismatch", len(bad), "rows", len(set(bad[:,0])) , np.abs(d).max())
