"""chunks_{model}.json <-> dense matrix.  Path/sanitising KATs are the reference's
own (rag_engine.rs:2366-2458); the file round trip and the re-normalise-on-load quirk run on the GPU."""
import json
import os

import numpy as np
import pytest

from conftest import bits


def test_sanitize_and_paths_kats(rlr, kats):
    for c in kats["sanitize_model_name"]:
        assert rlr.sanitize_model_name(c["in"]) == c["out"], c
    for c in kats["index_paths"]:
        assert rlr.get_index_path(c["data_dir"], c["model"]) == c["path"]
        assert rlr.get_index_path(c["data_dir"], c["model"]).startswith(c["data_dir"] + "/")
    assert rlr.get_legacy_path(kats["legacy_path"]["data_dir"]) == kats["legacy_path"]["path"]


def _write(path, state):
    with open(path, "w") as f:
        json.dump(state, f)


def _chunk(cid, doc, emb, idx=0, page=1):
    return {"id": cid, "document_name": doc, "text": f"text {cid}", "embedding": [float(x) for x in emb],
            "chunk_index": idx, "page_number": page, "section": None,
            "metadata": {"page_range": None, "sentence_range": None, "section_title": None, "token_count": 3,
                         "overlap_with_previous": 0}}


@pytest.mark.gpu
def test_load_renormalises_on_device_and_round_trips(rlr, oracle, tmp_path):
    dim, n, model = 768, 300, "nomic-embed-text"
    raw = (np.random.default_rng(3).standard_normal((n, dim)) * 2).astype(np.float32)  # NOT normalised on disk
    chunks = {f"id-{i}": _chunk(f"id-{i}", f"doc{i % 4}.pdf", raw[i], i) for i in range(n)}
    _write(rlr.get_index_path(str(tmp_path), model),
           {"version": 2, "model": model, "chunks": chunks, "needs_reindex": False,
            "document_hashes": {"doc0.pdf": "aa", "doc1.pdf": "bb", "doc2.pdf": "cc", "doc3.pdf": "dd", "gone.pdf": "zz"}})
    eng = rlr.RagEngine(dim)
    rep = rlr.load_from_disk(eng, str(tmp_path), model)
    assert rep.n_chunks == n and not rep.needs_reindex and not rep.migrated
    assert set(rep.document_hashes) == {"doc0.pdf", "doc1.pdf", "doc2.pdf", "doc3.pdf"}  # orphan dropped
    want = np.stack([oracle.normalize(r) for r in raw])       # rag_engine.rs:1678-1680
    got = eng.index.fetch_rows(np.arange(n))
    assert np.array_equal(bits(got), bits(want))
    q = oracle.synth_query(dim, seed=9)
    res = eng.search(q, 5)
    wr, wc, _, _ = oracle.search(want, q, 5)
    assert [r.row for r in res] == list(wr) and [r.chunk_id for r in res] == [f"id-{i}" for i in wr]
    assert np.array_equal(bits([r.score for r in res]), bits(wc))
    # save -> load again: values survive the decimal round trip, then get re-normalised once more
    path = rlr.save_to_disk(eng, str(tmp_path), model, document_hashes=rep.document_hashes)
    assert path.endswith("chunks_nomic-embed-text.json") and not os.path.exists(path[:-5] + ".json.tmp")
    state = json.load(open(path))
    assert state["version"] == 2 and state["model"] == model and len(state["chunks"]) == n
    on_disk = np.array([state["chunks"][f"id-{i}"]["embedding"] for i in range(n)], dtype=np.float64).astype(np.float32)
    assert np.array_equal(bits(on_disk), bits(got))            # shortest f32 literals round-trip exactly
    eng2 = rlr.RagEngine(dim)
    rep2 = rlr.load_from_disk(eng2, str(tmp_path), model)
    assert rep2.n_chunks == n
    again = np.stack([oracle.normalize(r) for r in got])       # the per-restart drift the survey notes
    assert np.array_equal(bits(eng2.index.fetch_rows(np.arange(n))), bits(again))
    eng.close(); eng2.close()


@pytest.mark.gpu
def test_sidecar_cache_gives_the_same_engine_and_detects_a_changed_file(rlr, oracle, tmp_path):
    dim, n, model = 256, 120, "cache-model"
    raw = (np.random.default_rng(5).standard_normal((n, dim)) * 3).astype(np.float32)
    chunks = {f"c{i}": _chunk(f"c{i}", f"d{i % 3}.pdf", raw[i], i, page=1 + i % 7) for i in range(n)}
    src = rlr.get_index_path(str(tmp_path), model)
    _write(src, {"version": 2, "model": model, "chunks": chunks, "needs_reindex": False,
                 "document_hashes": {"d0.pdf": "a", "d1.pdf": "b", "d2.pdf": "c"}})
    a = rlr.RagEngine(dim)
    rep_a = rlr.load_from_disk(a, str(tmp_path), model, use_sidecar=True)          # JSON load, cache written
    cache = rlr.get_sidecar_path(str(tmp_path), model)
    assert rep_a.source == src and os.path.exists(cache)
    b = rlr.RagEngine(dim)
    rep_b = rlr.load_from_disk(b, str(tmp_path), model, use_sidecar=True)          # served by the cache
    assert rep_b.source == cache and rep_b.n_chunks == n and rep_b.document_hashes == rep_a.document_hashes
    assert np.array_equal(bits(a.index.fetch_rows(np.arange(n))), bits(b.index.fetch_rows(np.arange(n))))
    q = oracle.synth_query(dim, seed=3)
    ra, rb = a.search(q, 7, query_text="text c5"), b.search(q, 7, query_text="text c5")
    assert [(x.chunk_id, x.page_number, x.document) for x in ra] == [(x.chunk_id, x.page_number, x.document) for x in rb]
    assert np.array_equal(bits([x.score for x in ra]), bits([x.score for x in rb]))
    # a changed JSON (other size / mtime) invalidates the cache: the loader goes back to the file and rewrites it
    chunks["c0"]["embedding"] = [float(x) for x in raw[1]]
    _write(src, {"version": 2, "model": model, "chunks": chunks, "needs_reindex": False, "document_hashes": {"d0.pdf": "a"}})
    os.utime(src, ns=(1, 1))                                                       # even with an OLDER timestamp
    c = rlr.RagEngine(dim)
    rep_c = rlr.load_from_disk(c, str(tmp_path), model, use_sidecar=True)
    assert rep_c.source == src
    assert np.array_equal(bits(c.index.fetch_rows([0])), bits(c.index.fetch_rows([1])))
    # a truncated cache is ignored, not trusted
    with open(cache, "r+b") as f:
        f.truncate(100)
    d = rlr.RagEngine(dim)
    assert rlr.load_from_disk(d, str(tmp_path), model, use_sidecar=True).source == src
    for e in (a, b, c, d):
        e.close()


@pytest.mark.gpu
def test_load_version1_clears_and_marks_reindex(rlr, tmp_path):
    model = "m"
    _write(rlr.get_index_path(str(tmp_path), model),
           {"version": 1, "model": model, "chunks": {"a": _chunk("a", "d.pdf", [1, 0, 0, 0])}, "needs_reindex": False})
    eng = rlr.RagEngine(4)
    rep = rlr.load_from_disk(eng, str(tmp_path), model)
    assert rep.needs_reindex and rep.n_chunks == 0 and len(eng) == 0
    state = json.load(open(rlr.get_index_path(str(tmp_path), model)))
    assert state["version"] == 2 and state["needs_reindex"] is True and state["chunks"] == {}
    eng.close()


@pytest.mark.gpu
def test_legacy_migration_and_model_mismatch(rlr, tmp_path):
    d = str(tmp_path)
    legacy = rlr.get_legacy_path(d)
    _write(legacy, {"version": 2, "model": "nomic-embed-text",
                    "chunks": {"migrated": _chunk("migrated", "doc.pdf", [3, 4, 0, 0])}, "needs_reindex": False,
                    "document_hashes": {"doc.pdf": "abc123"}})
    # another model: legacy preserved, start fresh (rag_engine.rs:1620-1628)
    eng = rlr.RagEngine(4)
    rep = rlr.load_from_disk(eng, d, "new-model")
    assert rep.n_chunks == 0 and rep.source is None and os.path.exists(legacy)
    assert not os.path.exists(rlr.get_index_path(d, "new-model"))
    # matching model: migrate to the model-specific file, legacy kept (:1597-1618, :1699-1706)
    rep = rlr.load_from_disk(eng, d, "nomic-embed-text")
    assert rep.migrated and rep.n_chunks == 1 and rep.document_hashes == {"doc.pdf": "abc123"}
    assert os.path.exists(legacy) and os.path.exists(rlr.get_index_path(d, "nomic-embed-text"))
    assert np.array_equal(eng.index.fetch_rows([0])[0], np.array([0.6, 0.8, 0, 0], np.float32))
    # corrupted model-specific file: kept, empty engine, reindex (:1571-1585)
    bad = rlr.get_index_path(d, "broken")
    open(bad, "w").write("{not json")
    eng3 = rlr.RagEngine(4)
    rep = rlr.load_from_disk(eng3, d, "broken")
    assert rep.needs_reindex and rep.n_chunks == 0 and os.path.exists(bad)
    eng.close(); eng3.close()
