"""The exchange step's merge kernel alone, world = 1 / 2 / 4 / 8 partial lists of k = 100: ms per rlr_merge_topk call (launch +
polled wait) and, under rocprofv3, the kernel's duration."""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
rlr = importlib.import_module("rust-local-rag_amd")
N = importlib.import_module("rust-local-rag_amd._native")
L = N.lib()
k, nq = 100, 1
rng = np.random.default_rng(0)
for world in (1, 2, 4, 8):
    lists = []
    for r in range(world):
        sc = np.sort(rng.random(k).astype(np.float32))[::-1]
        rows = rng.choice(1_000_000, size=k, replace=False).astype(np.uint32)
        lists.append([L.rlr_pack_result(C.c_float(float(s)), int(x)) for s, x in zip(sc, rows)])
    g = torch.tensor(np.array(lists, dtype=np.uint64).view(np.int64).reshape(world, nq, k), device="cuda")
    bases = (np.arange(world, dtype=np.uint64) * 1_250_000)
    rows_h, cos_h, n_h = np.zeros((nq, k), np.uint64), np.zeros((nq, k), np.float32), np.zeros(nq, np.uint32)
    stream = torch.cuda.current_stream().cuda_stream
    def call():
        N.check(L.rlr_merge_topk(0, C.c_void_p(g.data_ptr()), world, nq, k, bases.ctypes.data_as(N.u64p), rows_h.ctypes.data_as(N.u64p),
                                 cos_h.ctypes.data_as(N.f32p), n_h.ctypes.data_as(N.u32p), C.c_void_p(stream)))
    for _ in range(20):
        call()
    t0 = time.perf_counter()
    for _ in range(300):
        call()
    el = (time.perf_counter() - t0) / 300
    # check against numpy
    allk = np.array(lists, dtype=np.uint64).reshape(-1)
    print("world %d: %.1f us per call, n=%d" % (world, el * 1e6, int(n_h[0])), flush=True)
