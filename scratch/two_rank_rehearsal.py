"""Two ranks on ONE GPU (gloo carries the collectives, the tensors live on the GPU): the real multi-process data flow
of ShardedIndex -- begin/end search on torch's stream, all-gather of packed partials, merge kernel, winner-row
all-to-all + sharded MMR -- compared with a single index over the whole corpus.  The parent never touches the GPU.
usage: python scratch/two_rank_rehearsal.py"""
import importlib
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rlr = importlib.import_module("rust-local-rag_amd")
        sharded = importlib.import_module("rust-local-rag_amd.sharded")
        n, dim, k = 200_003, 768, 100
        sh = sharded.ShardedIndex(dim, n, "f32", device=0)
        sh.fill_synthetic(seed=77, n_clusters=300)
        rng = np.random.default_rng(5)
        qs = np.stack([rlr.normalize(rng.standard_normal(dim).astype(np.float32)) for _ in range(6)])
        out = {"single": [], "batch": None, "div": None}
        for q in qs[:3]:
            r, c = sh.search_topk(q, k)
            out["single"].append((r.copy(), c.copy()))
        out["batch"] = sh.search_topk(qs, 20)
        div = sh.search_with_diversity_batch(qs, 10, 0.7)
        out["div"] = [(a.copy(), b.copy()) for a, b, _ in div]
        if rank == 0:   # the whole corpus in one index, same process
            one = rlr.GpuIndex(dim)
            one.fill_synthetic(n, seed=77, n_clusters=300)
            ok = True
            for i in range(3):
                r1, c1 = one.search_topk(qs[i], k)
                ok &= bool(np.array_equal(r1[0].astype(np.int64), out["single"][i][0][0]) and
                           np.array_equal(c1[0].view(np.uint32), out["single"][i][1][0].view(np.uint32)))
            rb, cb = one.search_topk(qs, 20)
            ok &= bool(np.array_equal(rb.astype(np.int64), out["batch"][0]) and np.array_equal(cb.view(np.uint32), out["batch"][1].view(np.uint32)))
            pool = max(3 * 10, 10 + 10)
            rp, cp = one.search_topk(qs, pool)
            sc = (np.float32(0.7) * cp).astype(np.float32)
            o, _, cnt = one.mmr_select_batch(rp, sc, np.full(len(qs), pool, np.uint32), 10, 0.7)
            for i in range(len(qs)):
                want = rp[i][o[i, : cnt[i]].astype(np.int64)].astype(np.int64)
                ok &= bool(np.array_equal(want, out["div"][i][0]))
            ret["ok"] = ok
        ret[rank] = [(x[0].tolist()[:5]) for x, _ in [(s[0], s[1]) for s in out["single"]]]
    finally:
        dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, 2, port, ret)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    codes = [p.exitcode for p in procs]
    print("exit codes", codes, "| sharded == single index:", ret.get("ok"), "| ranks agree:", ret.get(0) == ret.get(1))
    sys.exit(0 if codes == [0, 0] and ret.get("ok") and ret.get(0) == ret.get(1) else 1)


if __name__ == "__main__":
    main()
