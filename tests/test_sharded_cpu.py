"""N>1 path on CPU: two processes, gloo, the same all-gather + merge code the GPU ranks run
(rust-local-rag_amd/sharded.py: gather_and_merge).  Each rank's "local search" is played by
the oracle over its own shard and packed with the library's rlr_pack_result, so what is under
test is the exchange format, the collective and the exact merge (tie rule included)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, dim, k, seed, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rlr = importlib.import_module("rust-local-rag_amd")
        sharded = importlib.import_module("rust-local-rag_amd.sharded")
        from oracle import oracle as O

        lo, hi = sharded.shard_range(n_total, rank, world)
        rows = O.synth_rows(hi - lo, dim, seed, row0=lo)
        if rank == 1:
            rows[3] = O.synth_rows(1, dim, seed, row0=5)[0]  # duplicate of global row 5 -> exact cross-shard tie
        qs = np.stack([O.normalize(O.synth_query(dim, seed + 1 + i)) for i in range(3)])
        L = rlr.lib()
        local = np.zeros((len(qs), k), dtype=np.uint64)
        for qi, q in enumerate(qs):
            e = O.scan(rows, q)
            order = np.lexsort((np.arange(len(e)), -e.astype(np.float64)))[:k]
            for j, r in enumerate(order):
                local[qi, j] = L.rlr_pack_result(float(e[r]), int(r))
        bases = torch.tensor([sharded.shard_range(n_total, r, world)[0] for r in range(world)], dtype=torch.int64)
        t_local = torch.from_numpy(local.view(np.int64))
        g_rows, g_key = sharded.gather_and_merge(t_local, bases, k, dist)
        ret[rank] = (g_rows.numpy().copy(), sharded.key_to_score(g_key.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total,world,k", [(1001, 2, 20), (50, 3, 20)])   # the second: shards shorter than k
def test_world2_gloo_allgather_merge_matches_global_oracle(oracle, n_total, world, k):
    import torch.multiprocessing as mp

    dim, seed = 64, 77
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, dim, k, seed, ret)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]

    sharded = importlib.import_module("rust-local-rag_amd.sharded")
    rows = oracle.synth_rows(n_total, dim, seed)
    lo1, _ = sharded.shard_range(n_total, 1, world)
    rows[lo1 + 3] = rows[5]
    for qi in range(3):
        q = oracle.normalize(oracle.synth_query(dim, seed + 1 + qi))
        e = oracle.scan(rows, q)
        order = np.lexsort((np.arange(n_total), -e.astype(np.float64)))[:k]
        for rank in range(world):
            g_rows, g_cos = ret[rank]
            assert np.array_equal(g_rows[qi], order), (rank, qi)
            assert np.array_equal(g_cos[qi].view(np.uint32), e[order].view(np.uint32))
    # the duplicated row ties exactly with global row 5; the lower global row must come first
    q0 = oracle.normalize(rows[5])
    e = oracle.scan(rows, q0)
    assert e[5] == e[lo1 + 3]


def test_shard_ranges_cover_everything():
    sharded = importlib.import_module("rust-local-rag_amd.sharded")
    for n, w in ((10_000_000, 8), (1001, 2), (7, 8), (0, 4), (100_000_000, 8)):
        spans = [sharded.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        per = (n + w - 1) // w
        assert all(hi - lo <= per for lo, hi in spans)


def test_merge_handles_short_shards_and_padding():
    import torch
    rlr = importlib.import_module("rust-local-rag_amd")
    sharded = importlib.import_module("rust-local-rag_amd.sharded")
    L = rlr.lib()
    k = 4
    # rank 0 has only 2 rows (rest zero padding), rank 1 has 4
    g = np.zeros((2, 1, k), dtype=np.uint64)
    g[0, 0, 0] = L.rlr_pack_result(0.5, 1)
    g[0, 0, 1] = L.rlr_pack_result(-0.25, 0)
    for j, (s, r) in enumerate(((0.9, 2), (0.5, 0), (0.1, 3), (-0.5, 1))):
        g[1, 0, j] = L.rlr_pack_result(s, r)
    bases = torch.tensor([0, 2], dtype=torch.int64)
    rows, key = sharded.merge_packed(torch.from_numpy(g.view(np.int64)), bases, k)
    assert rows[0].tolist() == [4, 1, 2, 5]          # 0.9@2+2, 0.5@1 (lower global row first), 0.5@0+2, 0.1@3+2
    assert sharded.key_to_score(key.numpy())[0].tolist() == [np.float32(0.9), 0.5, 0.5, np.float32(0.1)]
    rows6, _ = sharded.merge_packed(torch.from_numpy(g.view(np.int64)), bases, 8)
    assert rows6[0].tolist() == [4, 1, 2, 5, 0, 3, -1, -1]


# ---- MMR on sharded data: winner-row exchange (all-to-all) + pick-list all-gather ---------------
def _mmr_worker(rank, world, port, n_total, dim, nq, P, k, lam, seed, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sharded = importlib.import_module("rust-local-rag_amd.sharded")
        from oracle import oracle as O

        lo, hi = sharded.shard_range(n_total, rank, world)
        shard = O.synth_rows(hi - lo, dim, seed, row0=lo, n_clusters=5)
        bases = np.array([sharded.shard_range(n_total, r, world)[0] for r in range(world)], dtype=np.uint64)
        pools, scores, sizes = _pools(n_total, nq, P, seed)

        def fetch(local_rows):  # stands in for rlr_fetch_rows_device
            return torch.from_numpy(shard[local_rows.astype(np.int64)].reshape(-1, dim).copy())

        def mmr(values, sc, sz, kk, ll):  # stands in for rlr_mmr_select_values: the oracle, query by query
            v = values.numpy().reshape(len(sz), P, dim)
            order = np.zeros((len(sz), P), np.uint32); mm = np.zeros((len(sz), P), np.float32); n = np.zeros(len(sz), np.uint32)
            for j in range(len(sz)):
                o, m_ = O.mmr(v[j, : sz[j]], sc[j, : sz[j]], kk, ll)
                order[j, : len(o)] = o; mm[j, : len(o)] = m_; n[j] = len(o)
            return order, mm, n

        out = sharded.sharded_mmr(pools, scores, sizes, k, lam, rank=rank, world=world, bases=bases, dim=dim,
                                  fetch=fetch, mmr=mmr, torch=torch, dist=dist)
        ret[rank] = tuple(a.copy() for a in out)
    finally:
        dist.destroy_process_group()


def _pools(n_total, nq, P, seed):
    rng = np.random.default_rng(seed)
    pools = np.stack([rng.choice(n_total, size=P, replace=False) for _ in range(nq)]).astype(np.int64)
    scores = np.sort(rng.random((nq, P)).astype(np.float32), axis=1)[:, ::-1].copy()
    sizes = np.full(nq, P, dtype=np.uint32)
    sizes[1] = P - 3          # a ragged pool
    sizes[nq - 1] = 1         # a single-candidate pool
    return pools, scores, sizes


def test_world2_gloo_sharded_mmr_matches_global_oracle(oracle):
    import torch.multiprocessing as mp

    n_total, dim, nq, P, k, lam, seed, world = 301, 32, 5, 12, 6, 0.7, 91, 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_mmr_worker, args=(r, world, port, n_total, dim, nq, P, k, lam, seed, ret))
             for r in range(world)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]

    rows = oracle.synth_rows(n_total, dim, seed, n_clusters=5)
    pools, scores, sizes = _pools(n_total, nq, P, seed)
    for rank in range(world):
        order, mm, n = ret[rank]
        for q in range(nq):
            o, m_ = oracle.mmr(rows[pools[q, : sizes[q]]], scores[q, : sizes[q]], k, lam)
            assert n[q] == len(o), (rank, q)
            assert np.array_equal(order[q, : n[q]], o), (rank, q)
            assert np.array_equal(mm[q, 1: n[q]].view(np.uint32), m_[1:].view(np.uint32)), (rank, q)


def test_winner_exchange_plan_is_consistent_across_ranks():
    """every (src -> dst) count agrees on both sides and the permutation rebuilds the pool order"""
    sharded = importlib.import_module("rust-local-rag_amd.sharded")
    n_total, nq, P, world = 1000, 7, 9, 3
    pools, _, sizes = _pools(n_total, nq, P, 5)
    bases = np.array([sharded.shard_range(n_total, r, world)[0] for r in range(world)])
    plans = [sharded.plan_winner_exchange(pools, sizes, bases, r, world) for r in range(world)]
    for s in range(world):
        for d in range(world):
            assert plans[s][1][d] == plans[d][2][s]
    for d in range(world):
        # what dst receives, source by source
        recv = []
        for s in range(world):
            send_local, send_counts = plans[s][0], plans[s][1]
            off = int(send_counts[:d].sum())
            recv.extend((send_local[off: off + int(send_counts[d])].astype(np.int64) + bases[s]).tolist())
        perm, my_q = plans[d][3], plans[d][4]
        assert my_q.tolist() == list(range(d, nq, world))
        rebuilt = np.array(recv)[perm].reshape(len(my_q), P)
        for j, q in enumerate(my_q):
            assert np.array_equal(rebuilt[j, : sizes[q]], pools[q, : sizes[q]])


# ---- replicas-only mode: queries dealt round-robin, results all-gathered ---------------------------
class _OracleIndex:
    """stands in for GpuIndex on the CPU: exact top-k by the oracle"""

    def __init__(self, rows, O):
        self.rows, self.O = rows, O

    def __len__(self):
        return len(self.rows)

    def search_topk(self, qs, k):
        out_r, out_c = [], []
        for q in np.asarray(qs, dtype=np.float32).reshape(-1, self.rows.shape[1]):
            e = self.O.scan(self.rows, q)
            order = np.lexsort((np.arange(len(e)), -e.astype(np.float64)))[:k]
            out_r.append(order.astype(np.uint64)); out_c.append(e[order])
        return np.stack(out_r), np.stack(out_c)


def _replica_worker(rank, world, port, n, dim, k, seed, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sharded = importlib.import_module("rust-local-rag_amd.sharded")
        from oracle import oracle as O

        rows = O.synth_rows(n, dim, seed)
        qs = np.stack([O.normalize(O.synth_query(dim, seed + 1 + i)) for i in range(5)])
        rep = sharded.ReplicatedIndex(dim, index=_OracleIndex(rows, O), tensor_device="cpu")
        r, c = rep.search_topk(qs, k)
        ret[rank] = (r.copy(), c.copy())
    finally:
        dist.destroy_process_group()


def test_world2_gloo_replicated_queries(oracle):
    import torch.multiprocessing as mp

    n, dim, k, seed, world = 500, 48, 7, 123, 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_replica_worker, args=(r, world, port, n, dim, k, seed, ret)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    rows = oracle.synth_rows(n, dim, seed)
    want = _OracleIndex(rows, oracle).search_topk(
        np.stack([oracle.normalize(oracle.synth_query(dim, seed + 1 + i)) for i in range(5)]), k)
    for rank in range(world):
        r, c = ret[rank]
        assert np.array_equal(r.astype(np.uint64), want[0])
        assert np.array_equal(c.view(np.uint32), want[1].view(np.uint32))
