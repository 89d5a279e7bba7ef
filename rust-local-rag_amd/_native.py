"""ctypes view of librlr_gpu.so (include/rlr_gpu.h + include/rlr_engine.h + include/rlr_lexical.h).

The product path has no CPU implementation: if the HIP library is missing or cannot be
loaded this module raises, and every compute call on a box without a GPU returns
RLR_E_NO_DEVICE from the library itself.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "librlr_gpu.so")

RLR_OK = 0
RLR_E_INVALID, RLR_E_NO_DEVICE, RLR_E_HIP, RLR_E_OOM, RLR_E_RANGE, RLR_E_INTERNAL = -1, -2, -3, -4, -5, -6
RLR_F32, RLR_F16 = 0, 1
MAX_TOP_K = 100          # mcp_server.rs:364
DEFAULT_TOP_K = 5        # mcp_server.rs:85, :356-358
DEFAULT_DIVERSITY = 0.3  # mcp_server.rs:86, :359-361

f32p = C.POINTER(C.c_float)
u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)


class RlrError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"rlr status {status}: {message}")
        self.status = status


class QueryWeightsC(C.Structure):
    _fields_ = [("has_embedding", C.c_int32), ("embedding", C.c_float),
                ("has_lexical", C.c_int32), ("lexical", C.c_float),
                ("has_reranker", C.c_int32), ("reranker", C.c_float),
                ("has_initial", C.c_int32), ("initial", C.c_float)]


class ResolvedWeightsC(C.Structure):
    _fields_ = [("embedding", C.c_float), ("lexical", C.c_float),
                ("reranker", C.c_float), ("initial", C.c_float)]


class SearchHitC(C.Structure):
    _fields_ = [("row", C.c_uint64), ("score", C.c_float), ("embedding_score", C.c_float),
                ("lexical_score", C.c_float), ("initial_score", C.c_float)]


class JsonCorpusC(C.Structure):
    _fields_ = [("rows", C.POINTER(C.c_float)), ("n_rows", C.c_uint64), ("dim", C.c_uint32),
                ("meta_json", C.c_void_p), ("meta_len", C.c_uint64)]


class ProfileC(C.Structure):
    _fields_ = [("n_searches", C.c_uint64), ("n_scan_launches", C.c_uint64),
                ("scan_ms", C.c_double), ("select_ms", C.c_double), ("rescore_ms", C.c_double),
                ("total_ms", C.c_double), ("scan_bytes", C.c_uint64), ("n_candidates", C.c_uint64),
                ("n_retries", C.c_uint64),
                ("n_batches", C.c_uint64), ("n_batch_queries", C.c_uint64), ("batch_gemm_ms", C.c_double),
                ("batch_other_ms", C.c_double), ("batch_gemm_bytes", C.c_uint64), ("batch_gemm_flops", C.c_double),
                ("n_batch_fallbacks", C.c_uint64),
                ("batch_main_ms", C.c_double), ("batch_main_bytes", C.c_uint64), ("batch_main_flops", C.c_double),
                ("n_mmr", C.c_uint64), ("mmr_ms", C.c_double), ("n_batches_without_image", C.c_uint64)]


class MultiStatsC(C.Structure):
    _fields_ = [("n_topk_rccl", C.c_uint64), ("n_topk_host_merge", C.c_uint64), ("n_topk_rccl_fell_back", C.c_uint64),
                ("topk_rccl_ms", C.c_double), ("n_mmr_exchanges", C.c_uint64), ("mmr_exchange_bytes", C.c_uint64),
                ("mmr_exchange_ms", C.c_double), ("n_mmr_host_bounces", C.c_uint64)]


# every symbol include/*.h declares: (name, restype, argtypes)
_H = C.c_void_p
PROTOTYPES = [
    ("rlr_version", C.c_int32, []),
    ("rlr_device_count", C.c_int32, []),
    ("rlr_last_error", C.c_char_p, []),
    ("rlr_default_guard_eps", C.c_float, [C.c_uint32]),
    ("rlr_index_create", C.c_int32, [C.c_uint32, C.c_int32, C.c_int32, C.POINTER(_H)]),
    ("rlr_index_destroy", C.c_int32, [_H]),
    ("rlr_index_info", C.c_int32, [_H, u64p, u32p, i32p, i32p]),
    ("rlr_index_reserve", C.c_int32, [_H, C.c_uint64]),
    ("rlr_index_upload", C.c_int32, [_H, f32p, C.c_uint64, C.c_int32]),
    ("rlr_index_append", C.c_int32, [_H, f32p, C.c_uint64, C.c_int32, u64p]),
    ("rlr_index_delete_rows", C.c_int32, [_H, u64p, C.c_uint64]),
    ("rlr_index_fill_synthetic", C.c_int32, [_H, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]),
    ("rlr_index_enable_batch_image", C.c_int32, [_H, C.c_int32]),
    ("rlr_search_topk", C.c_int32, [_H, f32p, C.c_uint32, C.c_uint32, C.c_float, u64p, f32p, u32p]),
    ("rlr_search_topk_device", C.c_int32, [_H, f32p, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p]),
    ("rlr_search_topk_device_begin", C.c_int32, [_H, f32p, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p,
                                                 C.POINTER(C.c_void_p)]),
    ("rlr_search_topk_device_end", C.c_int32, [_H, C.c_void_p, u32p]),
    ("rlr_merge_topk", C.c_int32, [C.c_int32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, u64p, u64p, f32p, u32p,
                                   C.c_void_p]),
    ("rlr_pack_result", C.c_uint64, [C.c_float, C.c_uint32]),
    ("rlr_unpack_result", None, [C.c_uint64, f32p, u32p]),
    ("rlr_score_rows", C.c_int32, [_H, f32p, u64p, C.c_uint32, f32p]),
    ("rlr_fetch_rows", C.c_int32, [_H, u64p, C.c_uint32, f32p]),
    ("rlr_fetch_rows_device", C.c_int32, [_H, u64p, C.c_uint32, C.c_void_p]),
    ("rlr_mmr_select_values", C.c_int32, [_H, C.c_void_p, f32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, u32p,
                                          f32p, u32p]),
    ("rlr_mmr_select", C.c_int32, [_H, u64p, f32p, C.c_uint32, C.c_uint32, C.c_float, u32p, f32p, u32p]),
    ("rlr_mmr_select_batch", C.c_int32, [_H, u64p, f32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, u32p, f32p,
                                         u32p]),
    ("rlr_search_diverse", C.c_int32, [_H, f32p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float, u64p,
                                       f32p, f32p, u32p, i32p]),
    ("rlr_search_hybrid", C.c_int32, [_H, f32p, C.c_uint32, C.c_uint32, C.c_float, C.c_int32, C.c_float, C.c_float, u64p, f32p,
                                      C.c_uint32, C.c_float, C.c_float, u64p, f32p, f32p, f32p, u32p, i32p]),
    ("rlr_json_load_corpus", C.c_int32, [C.c_char_p, C.c_uint32, C.POINTER(JsonCorpusC)]),
    ("rlr_json_free_corpus", None, [C.POINTER(JsonCorpusC)]),
    ("rlr_index_load_json", C.c_int32, [_H, C.c_char_p, C.c_int32, C.POINTER(JsonCorpusC)]),
    ("rlr_json_format_embedding", C.c_uint64, [f32p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint64]),
    ("rlr_multi_create", C.c_int32, [C.c_uint32, C.c_int32, C.c_int32, i32p, C.POINTER(_H)]),
    ("rlr_multi_destroy", C.c_int32, [_H]),
    ("rlr_multi_info", C.c_int32, [_H, u64p, u32p]),
    ("rlr_multi_set_exchange", C.c_int32, [_H, C.c_int32]),
    ("rlr_multi_upload", C.c_int32, [_H, f32p, C.c_uint64, C.c_int32]),
    ("rlr_multi_fill_synthetic", C.c_int32, [_H, C.c_uint64, C.c_uint64, C.c_uint32]),
    ("rlr_multi_enable_batch_image", C.c_int32, [_H, C.c_int32]),
    ("rlr_multi_search_topk", C.c_int32, [_H, f32p, C.c_uint32, C.c_uint32, C.c_float, u64p, f32p, u32p]),
    ("rlr_multi_score_rows", C.c_int32, [_H, f32p, u64p, C.c_uint32, f32p]),
    ("rlr_multi_fetch_rows", C.c_int32, [_H, u64p, C.c_uint32, f32p]),
    ("rlr_multi_mmr_select", C.c_int32, [_H, u64p, f32p, C.c_uint32, C.c_uint32, C.c_float, u32p, f32p, u32p]),
    ("rlr_multi_mmr_select_batch", C.c_int32, [_H, u64p, f32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, u32p, f32p,
                                               u32p]),
    ("rlr_multi_stats", C.c_int32, [_H, C.POINTER(MultiStatsC), C.c_int32]),
    ("rlr_gather_rows_device", C.c_int32, [_H, u64p, C.c_uint32, C.c_void_p]),
    ("rlr_index_row_bytes", C.c_int32, [_H, u32p]),
    ("rlr_mmr_select_staged", C.c_int32, [_H, C.c_void_p, C.c_uint64, u64p, f32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_float, u32p, f32p, u32p]),
    ("rlr_profile_enable", C.c_int32, [_H, C.c_int32]),
    ("rlr_profile_read", C.c_int32, [_H, C.POINTER(ProfileC), C.c_int32]),
    ("rlr_index_probe_bandwidth", C.c_int32, [_H, C.c_int32, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    # rlr_engine.h
    ("rlr_resolve_weight", C.c_float, [C.c_int32, C.c_float, C.c_float]),
    ("rlr_resolve_weights", None, [C.POINTER(QueryWeightsC), C.POINTER(ResolvedWeightsC)]),
    ("rlr_normalize", None, [f32p, C.c_size_t]),
    ("rlr_engine_search", C.c_int32, [_H, f32p, C.c_uint32, C.c_uint32, C.POINTER(QueryWeightsC), u64p, f32p,
                                      C.c_uint32, C.c_int32, C.POINTER(SearchHitC), C.c_uint32, u32p]),
    ("rlr_engine_search_with_diversity", C.c_int32, [_H, f32p, C.c_uint32, C.c_uint32, C.c_float,
                                                     C.POINTER(QueryWeightsC), u64p, f32p, C.c_uint32,
                                                     C.POINTER(SearchHitC), C.c_uint32, u32p]),
    ("rlr_engine_search_text", C.c_int32, [_H, _H, f32p, C.c_uint32, C.c_char_p, C.c_size_t, C.c_uint32, C.c_float, C.c_int32,
                                           C.POINTER(QueryWeightsC), C.POINTER(SearchHitC), C.c_uint32, u32p]),
    ("rlr_engine_search_with_diversity_batch", C.c_int32, [_H, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float,
                                                           C.POINTER(QueryWeightsC), C.POINTER(SearchHitC), C.c_uint32,
                                                           u32p]),
    ("rlr_engine_blend_reranked", C.c_int32, [C.POINTER(SearchHitC), C.c_uint32, u64p, f32p, C.c_uint32, C.c_uint32,
                                              C.POINTER(QueryWeightsC), C.POINTER(SearchHitC), f32p, i32p, C.c_uint32,
                                              u32p]),
    ("rlr_engine_embedding_candidates", C.c_int32, [_H, f32p, C.c_uint32, C.c_uint32, u64p, f32p, u32p]),
    ("rlr_multi_engine_search", C.c_int32, [_H, f32p, C.c_uint32, C.c_uint32, C.POINTER(QueryWeightsC), u64p, f32p,
                                            C.c_uint32, C.c_int32, C.POINTER(SearchHitC), C.c_uint32, u32p]),
    ("rlr_multi_engine_search_with_diversity", C.c_int32, [_H, f32p, C.c_uint32, C.c_uint32, C.c_float,
                                                           C.POINTER(QueryWeightsC), u64p, f32p, C.c_uint32,
                                                           C.POINTER(SearchHitC), C.c_uint32, u32p]),
    ("rlr_multi_engine_search_text", C.c_int32, [_H, _H, f32p, C.c_uint32, C.c_char_p, C.c_size_t, C.c_uint32, C.c_float,
                                                 C.c_int32, C.POINTER(QueryWeightsC), C.POINTER(SearchHitC), C.c_uint32,
                                                 u32p]),
    ("rlr_multi_engine_search_with_diversity_batch", C.c_int32, [_H, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float,
                                                                 C.POINTER(QueryWeightsC), C.POINTER(SearchHitC),
                                                                 C.c_uint32, u32p]),
    ("rlr_multi_engine_embedding_candidates", C.c_int32, [_H, f32p, C.c_uint32, C.c_uint32, u64p, f32p, u32p]),
    # rlr_lexical.h
    ("rlr_lexical_create", C.c_int32, [C.c_int32, C.POINTER(C.c_void_p)]),
    ("rlr_lexical_destroy", None, [_H]),
    ("rlr_lexical_add_chunk", C.c_int32, [_H, C.c_uint64, C.c_char_p, C.c_size_t]),
    ("rlr_lexical_remove_rows", C.c_int32, [_H, u64p, C.c_uint32]),
    ("rlr_lexical_clear", C.c_int32, [_H]),
    ("rlr_lexical_contains", C.c_int32, [_H, C.c_uint64]),
    ("rlr_lexical_info", C.c_int32, [_H, u64p, u64p, u64p, u64p]),
    ("rlr_lexical_segments", C.c_int32, [_H, u64p, u64p, u64p, u64p, u64p]),
    ("rlr_lexical_score", C.c_int32, [_H, C.c_char_p, C.c_size_t, C.c_uint32, u64p, f32p, u32p]),
    ("rlr_tokenize_ascii", C.c_int32, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
]

_lib = None


def _preload_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP/HSA
    runtimes in one process cannot both own the device, so when torch is installed its copy is
    loaded first and librlr_gpu.so binds to it by SONAME -- whichever of torch / this package is
    imported first, the process ends up with one runtime.  Without torch, /opt/rocm's is used."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(p):
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib() -> C.CDLL:
    """Load librlr_gpu.so (built in-tree by build.py). Raises if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: build it with `python rust-local-rag_amd/build.py` "
                "(there is no CPU implementation of the search path to fall back to)")
        _preload_torch_hip_runtime()
        L = C.CDLL(SO_PATH)
        for name, res, args in PROTOTYPES:
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int) -> None:
    if status != RLR_OK:
        raise RlrError(status, lib().rlr_last_error().decode("utf-8", "replace"))
