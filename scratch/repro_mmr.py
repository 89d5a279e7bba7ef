import importlib, sys
import numpy as np
sys.path.insert(0, '.')
rlr = importlib.import_module("rust-local-rag_amd")
from oracle import oracle as O
def bits(a): return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
dim, dtype, n, seed = 1152, "f16", 17, 470119562
for (dim, dtype) in [(1152, "f16"), (1152, "f32"), (768, "f16"), (1024, "f16")]:
    rows = O.synth_rows(n, dim, seed=seed, n_clusters=0, f16=(dtype == "f16"))
    ix = rlr.GpuIndex(dim, dtype); ix.upload(rows)
    q = O.normalize(O.synth_query(dim, seed=seed + 7))
    r, c = ix.search_topk(q, 100)
    P = 17
    sc = (np.float32(0.7) * c[0][:P]).astype(np.float32)
    for lam in (1.0, 0.7, 0.0):
        for kk in (10, 17):
            o, m = ix.mmr_select(r[0][:P], sc, kk, lam)
            wo, wm = O.mmr(rows[r[0][:P].astype(np.int64)], sc, kk, lam)
            same_o = np.array_equal(o, wo); same_m = np.array_equal(bits(m[1:]), bits(wm[1:]))
            print(dim, dtype, "lam", lam, "k", kk, "order", same_o, "mmr", same_m)
            if not (same_o and same_m):
                print("  gpu ", o.tolist(), m.tolist()); print("  cpu ", wo.tolist(), wm.tolist())
                emb = rows[r[0][:P].astype(np.int64)]
                g = ix.fetch_rows(r[0][:P]); print("  rows equal:", np.array_equal(g.view(np.uint32), emb.view(np.uint32)))
    ix.close()
