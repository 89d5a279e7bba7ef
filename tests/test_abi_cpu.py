"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol the
headers declare, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, pf


def _declared_symbols():
    names = set()
    for hdr in ("rlr_gpu.h", "rlr_engine.h", "rlr_lexical.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(rlr_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol(rlr):
    L = C.CDLL(rlr.SO_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} is declared in include/*.h but not exported"
    # and the ctypes prototype table covers the same set
    from importlib import import_module
    native = import_module("rust-local-rag_amd._native")
    assert {n for n, _, _ in native.PROTOTYPES} == declared


def test_version_and_guard_eps(rlr):
    assert rlr.lib().rlr_version() == 121
    e768, e1024 = rlr.default_guard_eps(768), rlr.default_guard_eps(1024)
    # rigorous bound: (dim + depth) * 2^-24, i.e. ~5e-5 at 768-d (hard part 1)
    assert 768 * 2.0 ** -24 < e768 < 6e-5 and e768 < e1024 < 8e-5


def test_no_cpu_fallback_without_a_device(rlr, gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    with pytest.raises(rlr.RlrError) as ei:
        rlr.GpuIndex(768)
    assert ei.value.status == -2  # RLR_E_NO_DEVICE
    assert "no CPU path" in str(ei.value)


def test_pack_roundtrip_and_order(rlr):
    L = rlr.lib()
    vals = [1.0, 0.5, 0.0, -0.0, -0.25, float("inf"), float("-inf")]
    packed = [L.rlr_pack_result(v, 7) for v in vals]
    srt = sorted(zip(packed, vals), reverse=True)
    assert [v for _, v in srt][:3] == [float("inf"), 1.0, 0.5]
    assert srt[-1][1] == float("-inf")
    s, r = C.c_float(), C.c_uint32()
    L.rlr_unpack_result(L.rlr_pack_result(0.15625, 123456), C.byref(s), C.byref(r))
    assert (s.value, r.value) == (0.15625, 123456)
    # equal scores: the lower row packs higher (row asc under descending order)
    assert L.rlr_pack_result(0.5, 3) > L.rlr_pack_result(0.5, 4)
    # NaN orders last
    assert L.rlr_pack_result(float("nan"), 0) < L.rlr_pack_result(float("-inf"), 0xFFFFFFF0)


def test_host_normalize_matches_oracle_bitwise(rlr, oracle):
    rng = np.random.default_rng(1)
    for dim in (3, 384, 768, 1024):
        v = rng.standard_normal(dim).astype(np.float32)
        assert np.array_equal(rlr.normalize(v).view(np.uint32), oracle.normalize(v).view(np.uint32))
    tiny = np.full(8, 1e-12, dtype=np.float32)  # norm_sq <= 1e-20 -> unchanged
    assert np.array_equal(rlr.normalize(tiny), tiny)


def test_resolve_weight_kats(rlr, kats):
    for case in kats["resolve_weight"]:
        ov = None if case["override"] is None else pf(case["override"])
        got = rlr.resolve_weight(ov, case["default"])
        assert np.float32(got) == np.float32(pf(case["expect"])), case


def test_resolved_weights_kats(rlr, kats):
    for case in kats["resolved_weights"]:
        w = rlr.QueryWeights(**{k: pf(v) for k, v in case["weights"].items()})
        got = rlr.ResolvedWeights.from_query_weights(w)
        for k, v in case["expect"].items():
            assert np.float32(getattr(got, k)) == np.float32(v), (case["name"], k)
    d = rlr.ResolvedWeights.from_query_weights(None)
    assert (np.float32(d.embedding), np.float32(d.lexical)) == (np.float32(0.7), np.float32(0.3))


def test_format_search_results_kat(rlr, kats):
    case = kats["format_search_results"]
    res = [rlr.SearchResult(**r) for r in case["results"]]
    text = rlr.format_search_results(res)
    for needle in case["must_contain"]:
        assert needle in text
    assert rlr.format_search_results([]) == "No results found."


def test_search_result_json_skips_none_fields(rlr):
    r = rlr.SearchResult("t", 0.5, "d.pdf", "id", 0, 0)
    assert set(r.to_json()) == {"text", "score", "document", "chunk_id", "chunk_index", "page_number", "section"}
    r.embedding_score = 0.4
    assert "embedding_score" in r.to_json()


def test_integration_doc_shows_a_binding_for_every_symbol():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in sorted(_declared_symbols()) if n not in doc]
    assert not missing, f"INTEGRATION.md shows no reference-side binding for {missing}"
