"""GpuIndex: numpy-facing wrapper of the rlr_index C handle (include/rlr_gpu.h)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _native as N


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(N.f32p)


def _up(a: np.ndarray):
    return a.ctypes.data_as(N.u64p)


@dataclass
class Profile:
    n_searches: int
    n_scan_launches: int
    scan_ms: float
    select_ms: float
    rescore_ms: float
    total_ms: float
    scan_bytes: int
    n_candidates: int
    n_retries: int
    n_batches: int = 0
    n_batch_queries: int = 0
    batch_gemm_ms: float = 0.0
    batch_other_ms: float = 0.0
    batch_gemm_bytes: int = 0
    batch_gemm_flops: float = 0.0
    n_batch_fallbacks: int = 0
    batch_main_ms: float = 0.0
    batch_main_bytes: int = 0
    batch_main_flops: float = 0.0
    n_mmr: int = 0
    mmr_ms: float = 0.0
    n_batches_without_image: int = 0


class GpuIndex:
    """Dense chunk-embedding matrix resident in HBM + the search entry points."""

    def __init__(self, dim: int, dtype: str = "f32", device: int = 0):
        self._L = N.lib()
        self._h = C.c_void_p()
        code = {"f32": N.RLR_F32, "f16": N.RLR_F16}[dtype]
        N.check(self._L.rlr_index_create(dim, code, device, C.byref(self._h)))
        self.dim = dim
        self.dtype = dtype
        self.device = device

    # -- lifetime ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._L.rlr_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def __len__(self) -> int:
        n = C.c_uint64()
        N.check(self._L.rlr_index_info(self._h, C.byref(n), None, None, None))
        return int(n.value)

    # -- mutation -----------------------------------------------------------
    def upload(self, rows, normalize: bool = False) -> None:
        rows = _f32(rows).reshape(-1, self.dim) if np.size(rows) else np.zeros((0, self.dim), np.float32)
        N.check(self._L.rlr_index_upload(self._h, _fp(rows), rows.shape[0], int(normalize)))

    def append(self, rows, normalize: bool = False) -> int:
        rows = _f32(rows).reshape(-1, self.dim)
        first = C.c_uint64()
        N.check(self._L.rlr_index_append(self._h, _fp(rows), rows.shape[0], int(normalize), C.byref(first)))
        return int(first.value)

    def delete_rows(self, rows) -> None:
        rows = _u64(rows).ravel()
        if rows.size:
            N.check(self._L.rlr_index_delete_rows(self._h, _up(rows), rows.size))

    def reserve(self, n_rows: int) -> None:
        N.check(self._L.rlr_index_reserve(self._h, n_rows))

    def fill_synthetic(self, n_rows: int, seed: int, row0: int = 0, n_clusters: int = 0) -> None:
        N.check(self._L.rlr_index_fill_synthetic(self._h, n_rows, row0, seed, n_clusters))

    def enable_batch_image(self, on: bool = True, single_query: bool = False, q8: bool = False) -> None:
        """Optional nomination copies of the rows (results never change, only the bytes the scan streams):
        on            binary16 image for the batched matrix-core path (dim * 2 bytes per row)
        single_query  single queries over f32 rows nominate from that image too (half the scan bytes)
        q8            8-bit copy + per-row scale for single queries (a quarter of the scan bytes; dim * 1 + 4 B/row)"""
        flags = (1 if (on or single_query) else 0) | (2 if single_query else 0) | (4 if q8 else 0)
        N.check(self._L.rlr_index_enable_batch_image(self._h, flags))

    # -- hot path ------------------------------------------------------------
    def search_topk(self, queries, k: int, guard_eps: float = -1.0):
        """queries: [Q, dim] (or [dim]) already normalised -> (rows u64 [Q,k'], cos f32 [Q,k'])"""
        q = _f32(queries).reshape(-1, self.dim)
        nq = q.shape[0]
        rows = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        cos = np.zeros((nq, max(k, 1)), dtype=np.float32)
        n_out = np.zeros(max(nq, 1), dtype=np.uint32)
        N.check(self._L.rlr_search_topk(self._h, _fp(q), nq, k, guard_eps, _up(rows), _fp(cos),
                                        n_out.ctypes.data_as(N.u32p)))
        kk = int(n_out[0]) if nq else 0
        return rows[:, :kk], cos[:, :kk]

    def search_topk_device(self, queries, k: int, d_out_ptr: int, stream: int = 0, guard_eps: float = -1.0) -> None:
        q = _f32(queries).reshape(-1, self.dim)
        N.check(self._L.rlr_search_topk_device(self._h, _fp(q), q.shape[0], k, guard_eps,
                                               C.c_void_p(d_out_ptr), C.c_void_p(stream)))

    def search_topk_device_begin(self, queries, k: int, d_out_ptr: int, stream: int = 0, guard_eps: float = -1.0):
        """enqueue the search, make `stream` wait for it, return a ticket for search_topk_device_end"""
        q = _f32(queries).reshape(-1, self.dim)
        t = C.c_void_p()
        N.check(self._L.rlr_search_topk_device_begin(self._h, _fp(q), q.shape[0], k, guard_eps, C.c_void_p(d_out_ptr),
                                                     C.c_void_p(stream), C.byref(t)))
        return t

    def search_topk_device_end(self, ticket) -> int:
        """join; -> number of queries whose results are invalid (guard band overflow: redo synchronously)"""
        n = C.c_uint32()
        N.check(self._L.rlr_search_topk_device_end(self._h, ticket, C.byref(n)))
        return n.value

    def score_rows(self, query, rows) -> np.ndarray:
        q = _f32(query).ravel()
        rows = _u64(rows).ravel()
        out = np.zeros(rows.size, dtype=np.float32)
        if rows.size:
            N.check(self._L.rlr_score_rows(self._h, _fp(q), _up(rows), rows.size, _fp(out)))
        return out

    def fetch_rows(self, rows) -> np.ndarray:
        rows = _u64(rows).ravel()
        out = np.zeros((rows.size, self.dim), dtype=np.float32)
        if rows.size:
            N.check(self._L.rlr_fetch_rows(self._h, _up(rows), rows.size, _fp(out)))
        return out

    def mmr_select(self, pool_rows, pool_scores, k: int, lam: float):
        pool_rows = _u64(pool_rows).ravel()
        pool_scores = _f32(pool_scores).ravel()
        P = pool_rows.size
        order = np.zeros(max(P, 1), dtype=np.uint32)
        mmr = np.zeros(max(P, 1), dtype=np.float32)
        n = C.c_uint32()
        N.check(self._L.rlr_mmr_select(self._h, _up(pool_rows), _fp(pool_scores), P, k, lam,
                                       order.ctypes.data_as(N.u32p), _fp(mmr), C.byref(n)))
        return order[: n.value], mmr[: n.value]

    def mmr_select_batch(self, pool_rows, pool_scores, pool_sizes, k: int, lam: float):
        """pool_rows/pool_scores: [Q, P]; pool_sizes: [Q] -> (order u32 [Q, P], mmr f32 [Q, P], n u32 [Q])"""
        pool_rows = _u64(pool_rows)
        pool_scores = _f32(pool_scores)
        nq, P = pool_rows.shape
        sizes = np.ascontiguousarray(pool_sizes, dtype=np.uint32)
        order = np.zeros((nq, max(P, 1)), dtype=np.uint32)
        mmr = np.zeros((nq, max(P, 1)), dtype=np.float32)
        n = np.zeros(max(nq, 1), dtype=np.uint32)
        N.check(self._L.rlr_mmr_select_batch(self._h, _up(pool_rows), _fp(pool_scores), sizes.ctypes.data_as(N.u32p), nq, P,
                                             k, lam, order.ctypes.data_as(N.u32p), _fp(mmr), n.ctypes.data_as(N.u32p)))
        return order, mmr, n[:nq]

    def fetch_rows_device(self, rows, d_out_ptr: int) -> None:
        """rows of this index as dense f32 values into device memory at `d_out_ptr` (len(rows) x dim floats)"""
        rows = _u64(rows).ravel()
        if rows.size:
            N.check(self._L.rlr_fetch_rows_device(self._h, _up(rows), rows.size, C.c_void_p(d_out_ptr)))

    def mmr_select_values(self, d_values_ptr: int, pool_scores, pool_sizes, k: int, lam: float):
        """mmr_select_batch for pools whose row values are in device memory: [Q, P, dim] f32 at `d_values_ptr`"""
        pool_scores = _f32(pool_scores)
        nq, P = pool_scores.shape
        sizes = np.ascontiguousarray(pool_sizes, dtype=np.uint32)
        order = np.zeros((nq, max(P, 1)), dtype=np.uint32)
        mmr = np.zeros((nq, max(P, 1)), dtype=np.float32)
        n = np.zeros(max(nq, 1), dtype=np.uint32)
        N.check(self._L.rlr_mmr_select_values(self._h, C.c_void_p(d_values_ptr), _fp(pool_scores),
                                              sizes.ctypes.data_as(N.u32p), nq, P, k, lam,
                                              order.ctypes.data_as(N.u32p), _fp(mmr), n.ctypes.data_as(N.u32p)))
        return order, mmr, n[:nq]

    # -- measurement ---------------------------------------------------------
    def profile_enable(self, on: bool = True) -> None:
        N.check(self._L.rlr_profile_enable(self._h, int(on)))

    def profile_read(self, reset: bool = False) -> Profile:
        p = N.ProfileC()
        N.check(self._L.rlr_profile_read(self._h, C.byref(p), int(reset)))
        return Profile(*(getattr(p, f) for f, _ in N.ProfileC._fields_))

    def probe_bandwidth(self, mode: int, reps: int = 5):
        """measured-peak denominators over the index's own rows: mode 0 read-only stream, 1 device-to-device copy
        -> (GB/s, ms per launch)"""
        g, ms = C.c_double(), C.c_double()
        N.check(self._L.rlr_index_probe_bandwidth(self._h, mode, reps, C.byref(g), C.byref(ms)))
        return g.value, ms.value


class MultiGpuIndex:
    """One process, several GPUs (rlr_multi_*): contiguous row shards, host-side merge."""

    def __init__(self, dim: int, device_ids, dtype: str = "f32"):
        self._L = N.lib()
        self._h = C.c_void_p()
        ids = np.ascontiguousarray(device_ids, dtype=np.int32)
        code = {"f32": N.RLR_F32, "f16": N.RLR_F16}[dtype]
        N.check(self._L.rlr_multi_create(dim, code, ids.size, ids.ctypes.data_as(N.i32p), C.byref(self._h)))
        self.dim = dim

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._L.rlr_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        n = C.c_uint64()
        N.check(self._L.rlr_multi_info(self._h, C.byref(n), None))
        return int(n.value)

    def upload(self, rows, normalize: bool = False) -> None:
        rows = _f32(rows).reshape(-1, self.dim)
        N.check(self._L.rlr_multi_upload(self._h, _fp(rows), rows.shape[0], int(normalize)))

    def fill_synthetic(self, n_rows: int, seed: int, n_clusters: int = 0) -> None:
        N.check(self._L.rlr_multi_fill_synthetic(self._h, n_rows, seed, n_clusters))

    def enable_batch_image(self, flags: int = 1) -> None:
        """rlr_index_enable_batch_image on every shard (1: binary16 image for batches, 3: + single queries, 4: 8-bit copy)"""
        N.check(self._L.rlr_multi_enable_batch_image(self._h, flags))

    def set_exchange(self, mode: str) -> None:
        """"host": per-shard lists merged on the host (default); "rccl": ncclAllGather of the packed partial top-k
        lists + merge kernel on the first device (one shard per device; librccl loaded on first use)"""
        N.check(self._L.rlr_multi_set_exchange(self._h, {"host": 0, "rccl": 1}[mode]))

    def search_topk(self, queries, k: int, guard_eps: float = -1.0):
        q = _f32(queries).reshape(-1, self.dim)
        nq = q.shape[0]
        rows = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        cos = np.zeros((nq, max(k, 1)), dtype=np.float32)
        n_out = np.zeros(max(nq, 1), dtype=np.uint32)
        N.check(self._L.rlr_multi_search_topk(self._h, _fp(q), nq, k, guard_eps, _up(rows), _fp(cos),
                                              n_out.ctypes.data_as(N.u32p)))
        kk = int(n_out[0]) if nq else 0
        return rows[:, :kk], cos[:, :kk]

    def score_rows(self, query, rows) -> np.ndarray:
        q = _f32(query).ravel()
        rows = _u64(rows).ravel()
        out = np.zeros(rows.size, dtype=np.float32)
        if rows.size:
            N.check(self._L.rlr_multi_score_rows(self._h, _fp(q), _up(rows), rows.size, _fp(out)))
        return out

    def fetch_rows(self, rows) -> np.ndarray:
        rows = _u64(rows).ravel()
        out = np.zeros((rows.size, self.dim), dtype=np.float32)
        if rows.size:
            N.check(self._L.rlr_multi_fetch_rows(self._h, _up(rows), rows.size, _fp(out)))
        return out

    def mmr_select(self, pool_rows, pool_scores, k: int, lam: float):
        pool_rows = _u64(pool_rows).ravel()
        pool_scores = _f32(pool_scores).ravel()
        P = pool_rows.size
        order = np.zeros(max(P, 1), dtype=np.uint32)
        mmr = np.zeros(max(P, 1), dtype=np.float32)
        n = C.c_uint32()
        N.check(self._L.rlr_multi_mmr_select(self._h, _up(pool_rows), _fp(pool_scores), P, k, lam,
                                             order.ctypes.data_as(N.u32p), _fp(mmr), C.byref(n)))
        return order[: n.value], mmr[: n.value]

    def mmr_select_batch(self, pool_rows, pool_scores, pool_sizes, k: int, lam: float):
        """pools strided by P = pool_rows.shape[1]; -> (order [nq, P], mmr [nq, P], n_selected [nq])"""
        pool_rows = _u64(pool_rows)
        nq, P = pool_rows.shape
        pool_scores = _f32(pool_scores).reshape(nq, P)
        sizes = np.ascontiguousarray(pool_sizes, dtype=np.uint32)
        order = np.zeros((max(nq, 1), max(P, 1)), dtype=np.uint32)
        mmr = np.zeros((max(nq, 1), max(P, 1)), dtype=np.float32)
        n = np.zeros(max(nq, 1), dtype=np.uint32)
        N.check(self._L.rlr_multi_mmr_select_batch(self._h, _up(pool_rows), _fp(pool_scores), sizes.ctypes.data_as(N.u32p), nq, P, k,
                                                   lam, order.ctypes.data_as(N.u32p), _fp(mmr), n.ctypes.data_as(N.u32p)))
        return order[:nq], mmr[:nq], n[:nq]

    def stats(self, reset: bool = False) -> dict:
        st = N.MultiStatsC()
        N.check(self._L.rlr_multi_stats(self._h, C.byref(st), int(reset)))
        return {name: getattr(st, name) for name, _ in N.MultiStatsC._fields_}

    # ---- RagEngine::search / search_with_diversity over the sharded corpus (rlr_multi_engine_*) ----
    # results: numpy records (row, score, embedding_score, lexical_score, initial_score), global rows
    _HIT = np.dtype([("row", "<u8"), ("score", "<f4"), ("embedding_score", "<f4"), ("lexical_score", "<f4"),
                     ("initial_score", "<f4")])

    @staticmethod
    def _lex_args(lex_rows, lex_scores):
        if lex_rows is None or len(lex_rows) == 0:
            return np.zeros(1, np.uint64), np.zeros(1, np.float32), 0
        lr, ls = _u64(lex_rows).ravel(), _f32(lex_scores).ravel()
        return lr, ls, int(lr.size)

    def engine_search(self, query, top_k: int, weights=None, lex_rows=None, lex_scores=None, stage: int = 0):
        q = _f32(query).ravel()
        cap = max(3 * max(top_k, 1), 1)
        hits = (N.SearchHitC * cap)()
        n = C.c_uint32()
        lr, ls, nl = self._lex_args(lex_rows, lex_scores)
        wc = weights.to_c() if weights is not None else None
        N.check(self._L.rlr_multi_engine_search(self._h, _fp(q), q.size, top_k, C.byref(wc) if wc is not None else None, _up(lr),
                                                _fp(ls), nl, stage, hits, cap, C.byref(n)))
        return np.frombuffer(hits, dtype=self._HIT, count=n.value).copy()

    def engine_search_with_diversity(self, query, top_k: int, diversity_factor: float, weights=None, lex_rows=None,
                                     lex_scores=None):
        q = _f32(query).ravel()
        cap = max(3 * max(top_k, 1), top_k + 10)
        hits = (N.SearchHitC * cap)()
        n = C.c_uint32()
        lr, ls, nl = self._lex_args(lex_rows, lex_scores)
        wc = weights.to_c() if weights is not None else None
        N.check(self._L.rlr_multi_engine_search_with_diversity(self._h, _fp(q), q.size, top_k, float(diversity_factor),
                                                               C.byref(wc) if wc is not None else None, _up(lr), _fp(ls), nl,
                                                               hits, cap, C.byref(n)))
        return np.frombuffer(hits, dtype=self._HIT, count=n.value).copy()

    def engine_search_text(self, lexical, query, tokens: str, top_k: int, diversity_factor: float, weights=None, stage: int = 0):
        """lexical: a LexicalIndex over the global rows; tokens: the host's tokenize(query), space separated"""
        q = _f32(query).ravel()
        cap = max(3 * max(top_k, 1), top_k + 10)
        hits = (N.SearchHitC * cap)()
        n = C.c_uint32()
        tok = tokens.encode("utf-8")
        wc = weights.to_c() if weights is not None else None
        N.check(self._L.rlr_multi_engine_search_text(self._h, lexical._h, _fp(q), q.size, tok, len(tok), top_k,
                                                     float(diversity_factor), stage, C.byref(wc) if wc is not None else None,
                                                     hits, cap, C.byref(n)))
        return np.frombuffer(hits, dtype=self._HIT, count=n.value).copy()

    def engine_search_with_diversity_batch(self, queries, top_k: int, diversity_factor: float, weights=None):
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        nq, dq = q.shape
        cap = max(3 * max(top_k, 1), top_k + 10)
        hits = (N.SearchHitC * (cap * max(nq, 1)))()
        n_out = np.zeros(max(nq, 1), dtype=np.uint32)
        wc = weights.to_c() if weights is not None else None
        N.check(self._L.rlr_multi_engine_search_with_diversity_batch(self._h, _fp(q), dq, nq, top_k, float(diversity_factor),
                                                                     C.byref(wc) if wc is not None else None, hits, cap,
                                                                     n_out.ctypes.data_as(N.u32p)))
        allh = np.frombuffer(hits, dtype=self._HIT, count=cap * max(nq, 1)).reshape(max(nq, 1), cap)
        return [allh[i, : int(n_out[i])].copy() for i in range(nq)]

    def engine_embedding_candidates(self, query, count: int):
        q = _f32(query).ravel()
        rows = np.zeros(max(count, 1), dtype=np.uint64)
        sc = np.zeros(max(count, 1), dtype=np.float32)
        n = C.c_uint32()
        N.check(self._L.rlr_multi_engine_embedding_candidates(self._h, _fp(q), q.size, count, _up(rows), _fp(sc), C.byref(n)))
        return rows[: n.value], sc[: n.value]


def default_guard_eps(dim: int) -> float:
    return float(N.lib().rlr_default_guard_eps(dim))


def device_count() -> int:
    return int(N.lib().rlr_device_count())
