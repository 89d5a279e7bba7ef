"""Row-sharded search across the GPUs of one node: one process per GPU, one RCCL
all-gather of the per-shard partial top-k per query batch.

Each rank owns a contiguous row range of the corpus in its own GpuIndex.  A query runs
the full local pipeline (scan -> select -> reference-order re-score -> sort) on every
rank, the k best (score, local row) pairs stay in device memory in the packed u64 format
of rlr_search_topk_device, `torch.distributed.all_gather_into_tensor` moves world x k x 8
bytes over xGMI (3.2 KB for k=100 on 8 GPUs: latency-bound, never the per-link
bandwidth), and every rank merges the gathered lists.  Local scores are already the
reference-order values, so the merge is exact: global order = (score desc, global row
asc) -- shards are ascending row ranges, so (rank, local row) order is global row order.

torch is used for device memory, the process group and the merge's sort only.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

import ctypes as C
import os

from . import _native as N
from .index import GpuIndex, _f32


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """rows [lo, hi) of rank `rank`: ceil(N/G)-sized contiguous blocks."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def merge_packed(gathered, bases, k: int):
    """gathered: int64 tensor [world, Q, k] of packed results; bases: int64 tensor [world].
    Returns (global_rows int64 [Q, k'], score_bits int64 [Q, k']) sorted (score desc, row asc)."""
    import torch

    world, nq, kk = gathered.shape
    key = (gathered >> 32) & 0xFFFFFFFF                     # ordered score bits
    local = 0xFFFFFFFF - (gathered & 0xFFFFFFFF)
    valid = gathered != 0
    glob = local + bases.view(world, 1, 1)
    # one descending int64 per candidate: score key, then lower global row first
    comp = (key << 31) | ((1 << 31) - 1 - glob)
    comp = torch.where(valid, comp, torch.full_like(comp, -1))
    comp = comp.permute(1, 0, 2).reshape(nq, world * kk)
    top = torch.topk(comp, min(k, world * kk), dim=1, largest=True, sorted=True).values
    ok = top >= 0
    rows = torch.where(ok, (1 << 31) - 1 - (top & ((1 << 31) - 1)), torch.full_like(top, -1))
    skey = torch.where(ok, top >> 31, torch.zeros_like(top))
    return rows, skey


def key_to_score(skey: np.ndarray) -> np.ndarray:
    """inverse of csrc/common.h score_key for the merged keys (host, numpy)."""
    k = skey.astype(np.uint64).astype(np.uint32)
    b = np.where(k & np.uint32(0x80000000), k & np.uint32(0x7FFFFFFF), ~k)
    out = b.astype(np.uint32).view(np.float32).copy()
    out[k == 0] = np.nan
    return out


def gather_and_merge(local_packed, bases, k: int, dist=None, group=None, out=None):
    """The exchange step: all-gather every rank's packed [Q, k] partial top-k (RCCL on GPU
    tensors, gloo on CPU tensors in the tests) and merge.  Returns merge_packed's result."""
    import torch

    world = bases.numel()
    nq, kk = local_packed.shape
    if world > 1:
        if out is None:
            out = torch.zeros((world, nq, kk), dtype=torch.int64, device=local_packed.device)
        # rank-major concatenation along dim 0: the layout both RCCL and gloo accept
        dist.all_gather_into_tensor(out.view(world * nq, kk), local_packed.contiguous(), group=group)
        gathered = out
    else:
        gathered = local_packed.view(1, nq, kk)
    return merge_packed(gathered, bases, k)


class ShardedIndex:
    """One shard per process; collective search over a torch.distributed process group."""

    def __init__(self, dim: int, n_total: int, dtype: str = "f32", device: int = 0, group=None,
                 rank: Optional[int] = None, world: Optional[int] = None, index: Optional[GpuIndex] = None):
        import torch
        import torch.distributed as dist

        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = group
        self.rank = rank if rank is not None else (self.dist.get_rank(group) if self.dist else 0)
        self.world = world if world is not None else (self.dist.get_world_size(group) if self.dist else 1)
        self.n_total = n_total
        self.lo, self.hi = shard_range(n_total, self.rank, self.world)
        if n_total >= (1 << 32):
            raise ValueError("the merge key packs the global row into 32 bits")
        self.index = index if index is not None else GpuIndex(dim, dtype, device)
        self.dim = dim
        self.torch = torch
        self.dev = torch.device("cuda", device)
        self.bases = torch.tensor([shard_range(n_total, r, self.world)[0] for r in range(self.world)],
                                  dtype=torch.int64, device=self.dev)
        self._bases_h = np.array([shard_range(n_total, r, self.world)[0] for r in range(self.world)], dtype=np.uint64)
        self._local = None
        self._gath = None
        self._stream0 = None
        self._merge_args = None
        self._time_exchange = False
        self._exchange_ms = 0.0
        self._exchange_n = 0
        self._ev = None

    def time_exchange(self, on: bool) -> None:
        """Record torch events around the exchange step (all-gather + merge kernel) on torch's current stream,
        the stream that step is enqueued on; read the mean with exchange_ms_per_step()."""
        self._time_exchange = bool(on)
        if on:
            self._exchange_ms, self._exchange_n = 0.0, 0
            if self._ev is None:
                self._ev = (self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True))

    def exchange_ms_per_step(self):
        return self._exchange_ms / self._exchange_n if self._exchange_n else None

    def fill_synthetic(self, seed: int, n_clusters: int = 0) -> None:
        self.index.fill_synthetic(self.hi - self.lo, seed, row0=self.lo, n_clusters=n_clusters)

    def upload_shard(self, rows_global) -> None:
        self.index.upload(_f32(rows_global)[self.lo:self.hi])

    def search_topk(self, queries, k: int, _redo: bool = False):
        """-> (global rows int64 [Q,k'], cos f32 [Q,k']) identical on every rank."""
        torch = self.torch
        q = _f32(queries).reshape(-1, self.dim)
        nq = q.shape[0]
        if self._local is None or self._local.shape != (nq, k):
            self._local = torch.zeros((nq, k), dtype=torch.int64, device=self.dev)
            self._gath = torch.zeros((self.world, nq, k), dtype=torch.int64, device=self.dev)
            self._rows_h = np.zeros((nq, k), dtype=np.uint64)
            self._cos_h = np.zeros((nq, k), dtype=np.float32)
            self._n_h = np.zeros(nq, dtype=np.uint32)
            self._local_ptr = self._local.data_ptr()   # (torch's accessors cost ~1 us a call: looked up once per buffer)
            self._gath_ptr = self._gath.data_ptr()
        # torch's current stream carries the collective; without one (world 1) the library's calls only need A stream
        if self.world > 1 or self.dist is not None or self._stream0 is None:
            stream = self._stream0 = torch.cuda.current_stream(self.dev).cuda_stream
        else:
            stream = self._stream0
        if _redo:
            self.index.search_topk_device(q, k, self._local_ptr, stream)   # synchronous, handles overflow
            ticket = None
        else:
            # the collective and the merge are queued behind the scan: no host round trip in between
            ticket = self.index.search_topk_device_begin(q, k, self._local_ptr, stream)
        rows_h, cos_h, n_h = self._rows_h, self._cos_h, self._n_h
        try:
            if self._time_exchange:
                self._ev[0].record()
            if self.world > 1 or (self.dist is not None and os.environ.get("RLR_BENCH_FORCE_DIST") == "1"):
                # the exchange step: world x k x 8 B per query over xGMI (RCCL all-gather)
                self.dist.all_gather_into_tensor(self._gath.view(self.world * nq, k), self._local, group=self.group)
                gathered_ptr = self._gath_ptr
            else:
                gathered_ptr = self._local_ptr
            # merge on the GPU in one launch (rlr_merge_topk), results land in pinned host memory
            ma = self._merge_args
            if ma is None or ma[0] is not rows_h:   # (ctypes views of the persistent buffers: ~1 us each to build)
                ma = self._merge_args = (rows_h, self._bases_h.ctypes.data_as(N.u64p), rows_h.ctypes.data_as(N.u64p),
                                         cos_h.ctypes.data_as(N.f32p), n_h.ctypes.data_as(N.u32p), N.lib().rlr_merge_topk)
            N.check(ma[5](self.dev.index, C.c_void_p(gathered_ptr), self.world, nq, k, ma[1], ma[2], ma[3], ma[4],
                          C.c_void_p(stream)))
            if self._time_exchange:
                self._ev[1].record()
        finally:
            # the ticket holds a search context (pinned + device buffers): always hand it back, also when the
            # collective or the merge raised
            if ticket is not None:
                self.index.search_topk_device_end(ticket)
        if self._time_exchange:
            self._ev[1].synchronize()
            self._exchange_ms += self._ev[0].elapsed_time(self._ev[1])
            self._exchange_n += 1
        # A shard whose guard band overflowed (massive exact ties) marks its slot; the marker travels through the
        # all-gather, so every rank sees it in the merged counts and takes the same branch: redo the step on
        # the synchronous path, which handles the overflow.
        if not _redo and nq and (int(n_h[0]) == 0xFFFFFFFF if nq == 1 else bool((n_h[:nq] == 0xFFFFFFFF).any())):
            return self.search_topk(queries, k, _redo=True)
        n_valid = int(n_h[0]) if nq else 0
        return rows_h[:, :n_valid].astype(np.int64), cos_h[:, :n_valid].copy()

    # -- MMR on sharded data --------------------------------------------
    def mmr_select_batch(self, pool_rows, pool_scores, pool_sizes, k: int, lam: float):
        """Batched `mmr_diversify` over pools of GLOBAL rows (identical on every rank, as search_topk
        returns them).  -> (order u32 [Q, P], mmr f32 [Q, P], n u32 [Q]) on every rank."""
        torch = self.torch

        def fetch(local_rows: np.ndarray):
            out = torch.empty((max(local_rows.size, 1), self.dim), dtype=torch.float32, device=self.dev)
            self.index.fetch_rows_device(local_rows, out.data_ptr())
            return out[: local_rows.size]

        def mmr(values, scores, sizes, kk, ll):
            torch.cuda.current_stream(self.dev).synchronize()  # the exchange ran on torch's stream
            return self.index.mmr_select_values(values.data_ptr(), scores, sizes, kk, ll)

        return sharded_mmr(pool_rows, pool_scores, pool_sizes, k, lam, rank=self.rank, world=self.world,
                           bases=self._bases_h, dim=self.dim, fetch=fetch, mmr=mmr, torch=torch,
                           dist=self.dist if self.world > 1 else None, group=self.group, device=self.dev)

    def search_with_diversity_batch(self, queries, k: int, diversity: float, w_e: float = 0.7):
        """`search_with_diversity` (rag_engine.rs:717-759) for a batch of normalised queries over the
        sharded corpus, no lexical term: -> list of (global rows in pick order, relevance scores, cosines)."""
        lam = min(max(float(diversity), 0.0), 1.0)                 # :725
        w = np.float32(w_e)
        if lam == 0.0:                                             # :728-730
            rows, cos = self.search_topk(queries, k)
            return [(rows[q], (w * cos[q]).astype(np.float32), cos[q]) for q in range(rows.shape[0])]
        pool = max(3 * k, k + 10)                                  # :734
        rows, cos = self.search_topk(queries, pool)                # :735
        nq, P = rows.shape
        scores = (w * cos).astype(np.float32)                      # combined score, lexical part absent (:531-532)
        sizes = np.full(nq, P, dtype=np.uint32)
        order, _, n = self.mmr_select_batch(rows, scores, sizes, k, lam)  # :756
        out = []
        for q in range(nq):
            o = order[q, : n[q]].astype(np.int64)
            out.append((rows[q][o], scores[q][o], cos[q][o]))
        return out


def plan_winner_exchange(pool_rows, pool_sizes, bases, rank: int, world: int):
    """Who sends which pool rows where.  Query q is diversified on rank q % world; the owner of a
    pool row is the shard whose range holds it.  Every rank computes the same plan from the
    (replicated) merged pools, so no size exchange is needed.
      -> send_local   local rows this rank gathers, grouped by destination in (dst, query, position) order
         send_counts  [world] rows per destination
         recv_counts  [world] rows arriving per source
         perm         [m * P] slot (j-th own query, position p) -> index into the receive buffer
                      (sources concatenated in rank order); unused slots point at 0
         my_queries   the queries this rank diversifies"""
    rows = np.asarray(pool_rows, dtype=np.int64)
    nq, P = rows.shape
    sizes = np.asarray(pool_sizes, dtype=np.int64)
    bases = np.asarray(bases, dtype=np.int64)
    valid = np.arange(P)[None, :] < sizes[:, None]
    owner = np.searchsorted(bases, rows, side="right") - 1
    owner = np.where(valid, owner, -1)
    dst = np.broadcast_to((np.arange(nq) % world)[:, None], (nq, P))
    # send side: my rows, stable-sorted by destination (row-major flattening is (query, position) order)
    mine = (owner == rank).ravel()
    d_m = dst.ravel()[mine]
    order = np.argsort(d_m, kind="stable")
    send_local = (rows.ravel()[mine] - bases[rank])[order].astype(np.uint64)
    send_counts = np.bincount(d_m, minlength=world).astype(np.int64)
    # receive side: slots of my queries, arrival order = (source, query, position)
    my_queries = np.arange(rank, nq, world)
    own = owner[my_queries].ravel()
    ok = own >= 0
    arrival = np.argsort(own[ok], kind="stable")
    recv_counts = np.bincount(own[ok], minlength=world).astype(np.int64)
    perm = np.zeros(own.size, dtype=np.int64)
    slots = np.flatnonzero(ok)
    perm[slots[arrival]] = np.arange(arrival.size)
    return send_local, send_counts, recv_counts, perm, my_queries


def sharded_mmr(pool_rows, pool_scores, pool_sizes, k: int, lam: float, *, rank: int, world: int, bases, dim: int,
                fetch, mmr, torch, dist=None, group=None, device=None):
    """MMR over pools whose rows live on different ranks: winner-row exchange (one all-to-all of
    exactly the needed rows), local batched MMR for the queries q % world == rank, one all-gather of
    the pick lists.  `fetch(local_rows) -> tensor [n, dim] f32` and `mmr(values, scores, sizes, k, lam)
    -> (order, mmr, n)` are the shard's row gather and batched MMR (GPU entry points in ShardedIndex,
    CPU stand-ins in the gloo test)."""
    rows = np.asarray(pool_rows, dtype=np.int64)
    nq, P = rows.shape
    scores = np.ascontiguousarray(pool_scores, dtype=np.float32)
    sizes = np.ascontiguousarray(pool_sizes, dtype=np.uint32)
    send_local, send_counts, recv_counts, perm, my_q = plan_winner_exchange(rows, sizes, bases, rank, world)
    send = fetch(send_local)
    if dist is not None and world > 1:
        recv = torch.empty((max(int(recv_counts.sum()), 1), dim), dtype=torch.float32, device=send.device)
        n_recv = int(recv_counts.sum())
        dist.all_to_all_single(recv[:n_recv], send.contiguous(), output_split_sizes=[int(c) for c in recv_counts],
                               input_split_sizes=[int(c) for c in send_counts], group=group)
    else:
        recv = send
    m = my_q.size
    per = (nq + world - 1) // world  # padded own-query count, equal on all ranks
    res = np.zeros((per, 2 * P + 1), dtype=np.int32)
    if m and P:
        if recv.shape[0] == 0:
            recv = torch.zeros((1, dim), dtype=torch.float32, device=send.device)
        values = recv.index_select(0, torch.from_numpy(perm).to(recv.device)).contiguous()
        order, mm, n = mmr(values, scores[my_q], sizes[my_q], k, lam)
        res[:m, :P] = order.view(np.int32)
        res[:m, P:2 * P] = mm.view(np.int32)
        res[:m, 2 * P] = n.astype(np.int32)
    if dist is not None and world > 1:
        mine = torch.from_numpy(res).to(send.device)
        allr = torch.empty((world * per, 2 * P + 1), dtype=torch.int32, device=send.device)
        dist.all_gather_into_tensor(allr, mine, group=group)
        allr = allr.cpu().numpy().reshape(world, per, 2 * P + 1)
    else:
        allr = res.reshape(1, per, 2 * P + 1)
    order_out = np.zeros((nq, max(P, 1)), dtype=np.uint32)
    mmr_out = np.zeros((nq, max(P, 1)), dtype=np.float32)
    n_out = np.zeros(nq, dtype=np.uint32)
    for r in range(world):
        qs = np.arange(r, nq, world)
        if qs.size == 0 or P == 0:
            continue
        blk = allr[r, : qs.size]
        order_out[qs, :P] = blk[:, :P].view(np.uint32)
        mmr_out[qs, :P] = np.ascontiguousarray(blk[:, P:2 * P]).view(np.float32)
        n_out[qs] = blk[:, 2 * P].astype(np.uint32)
    return order_out, mmr_out, n_out


class ReplicatedIndex:
    """Replicas-only mode (last bullet): the corpus fits one GPU, every rank holds a
    full copy and the QUERIES are sharded -- rank r answers queries r, r + world, ... -- so a batch of
    queries costs no data-path collective for the search itself; one all-gather returns everybody's
    results to every rank (skip it with `gather=False` when each rank only needs its own answers)."""

    def __init__(self, dim: int, dtype: str = "f32", device: int = 0, group=None, rank: Optional[int] = None,
                 world: Optional[int] = None, index: Optional[GpuIndex] = None, tensor_device=None):
        import torch
        import torch.distributed as dist

        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = group
        self.rank = rank if rank is not None else (self.dist.get_rank(group) if self.dist else 0)
        self.world = world if world is not None else (self.dist.get_world_size(group) if self.dist else 1)
        self.index = index if index is not None else GpuIndex(dim, dtype, device)
        self.dim = dim
        self.torch = torch
        # tensors handed to the collective: the GPU for RCCL; the gloo test passes "cpu"
        self.dev = torch.device(tensor_device) if tensor_device is not None else torch.device("cuda", device)

    def my_queries(self, nq: int) -> np.ndarray:
        return np.arange(self.rank, nq, self.world)

    def search_topk(self, queries, k: int, gather: bool = True):
        """-> (rows int64 [Q, k'], cos f32 [Q, k']); with gather=False only this rank's queries
        (`my_queries(Q)`) are filled in, the others are -1 / NaN."""
        q = _f32(queries).reshape(-1, self.dim)
        nq = q.shape[0]
        kk = min(k, len(self.index))
        rows = np.full((nq, kk), -1, dtype=np.int64)
        cos = np.full((nq, kk), np.nan, dtype=np.float32)
        mine = self.my_queries(nq)
        if mine.size and kk:
            r, c = self.index.search_topk(q[mine], k)
            rows[mine], cos[mine] = r.astype(np.int64), c
        if gather and self.world > 1 and self.dist is not None and kk:
            torch = self.torch
            per = (nq + self.world - 1) // self.world
            buf = np.zeros((per, 2 * kk), dtype=np.int64)
            buf[: mine.size, :kk] = rows[mine]
            buf[: mine.size, kk:] = cos[mine].view(np.uint32).astype(np.int64)
            out = torch.empty((self.world * per, 2 * kk), dtype=torch.int64, device=self.dev)
            self.dist.all_gather_into_tensor(out, torch.from_numpy(buf).to(self.dev), group=self.group)
            allr = out.cpu().numpy().reshape(self.world, per, 2 * kk)
            for r in range(self.world):
                qs = np.arange(r, nq, self.world)
                rows[qs] = allr[r, : qs.size, :kk]
                cos[qs] = allr[r, : qs.size, kk:].astype(np.uint32).view(np.float32)
        return rows, cos
