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


# ---------------------------------------------------------------- f1 at scale: streaming reader / writer (host code)
def _native_read(rlr, path, dim):
    import ctypes as C
    N = rlr._native
    c = N.JsonCorpusC()
    N.check(N.lib().rlr_json_load_corpus(os.fsencode(path), dim, C.byref(c)))
    n = int(c.n_rows)
    rows = np.ctypeslib.as_array(c.rows, shape=(max(n * dim, 1),))[: n * dim].reshape(n, dim).copy()
    meta = C.string_at(c.meta_json, c.meta_len).decode("utf-8")
    N.lib().rlr_json_free_corpus(C.byref(c))
    return rows, meta


def test_streaming_reader_equals_the_json_module_on_hostile_documents(rlr, tmp_path):
    """rlr_json_load_corpus against json.load + float64 -> float32 (what serde_json's f32 path does): strings that
    contain the key it looks for, escapes, exponents, integers, null, short / long / missing / repeated embeddings,
    nested metadata, compact and pretty layouts -- rows bit-identical, metadata document = the file with every
    embedding array replaced by []."""
    persistence = __import__("importlib").import_module("rust-local-rag_amd.persistence")
    dim = 6
    chunks = {
        "a": {"id": "a", "text": 'he said "embedding": [9, 9, 9] \\" and left', "embedding": [1, -2.5, 3e-3, 4E+2, -0.0, 1e-50],
              "metadata": {"page_range": [1, 2], "nested": {"embedding": [7, 7]}}},
        "b\"q": {"embedding": [0.1, 0.2], "text": "short row é中", "document_name": "d.pdf"},
        "c": {"text": "long row", "embedding": [1, 2, 3, 4, 5, 6, 7, 8, 9]},
        "d": {"text": "no embedding at all"},
        "e": {"embedding": [], "text": "empty"},
        "f": {"embedding": [1.0000001, 16777217, 3.4028235e38, 1e39, -1e39, 1.401298464324817e-45], "text": "edges"},
        "g": {"text": "null inside", "embedding": [None, 2, None]},
    }
    state = {"version": 2, "model": "m", "document_hashes": {"d.pdf": "h"}, "chunks": chunks, "needs_reindex": False,
             "embedding": [5, 5], "trailer": {"chunks": {"x": {"embedding": [1]}}}}
    for name, kw in (("compact.json", dict(separators=(",", ":"))), ("pretty.json", dict(indent=2)),
                     ("ascii.json", dict(indent=1, ensure_ascii=True))):
        path = str(tmp_path / name)
        with open(path, "w", encoding="utf-8") as f:
            json.dump(state, f, ensure_ascii=kw.pop("ensure_ascii", False), **kw)
        rows, meta = _native_read(rlr, path, dim)
        want_state, want_rows = persistence._read_state(path, dim, native=False)
        assert rows.shape == (len(chunks), dim)
        with np.errstate(over="ignore"):
            assert np.array_equal(bits(rows), bits(want_rows)), name
        got_state = json.loads(meta)
        stripped = json.loads(json.dumps(state))
        for c in stripped["chunks"].values():
            if "embedding" in c:
                c["embedding"] = []
        assert got_state == stripped, name                      # top-level "embedding" and the trailer stay untouched
        assert list(got_state["chunks"].keys()) == list(chunks.keys())
    # a repeated key: the last one wins, as in every JSON map
    path = str(tmp_path / "dup.json")
    open(path, "w").write('{"chunks": {"k": {"embedding": [1, 1], "embedding": [2]}}, "version": 2, "model": "m"}')
    rows, meta = _native_read(rlr, path, 3)
    assert rows.tolist() == [[2.0, 0.0, 0.0]]
    # the document is a MAP (serde_json HashMap, Python dict): a repeated chunk id keeps its first position and takes
    # the last value, two spellings of one id are one id, a repeated "chunks" member starts over -- rows must stay
    # paired with the chunks json.loads sees
    dup_docs = [
        '{"version": 2, "chunks": {"a": {"embedding": [1, 1]}, "b": {"embedding": [2]}, "a": {"embedding": [3]}, "c": {"embedding": [4]}}}',
        '{"version": 2, "chunks": {"a": {"embedding": [1, 1]}, "b": {"embedding": [2]}, "\\u0061": {"text": "no embedding now"}}}',
        '{"version": 2, "chunks": {"a": {"embedding": [1]}, "a": 5, "b": {"embedding": [2]}}}',
        '{"chunks": {"x": {"embedding": [9]}, "y": {"embedding": [8]}}, "version": 2, "chunks": {"p": {"embedding": [1, 2, 3]}}}',
        '{"chunks": {"x": {"embedding": [9]}}, "ch\\u0075nks": {"q": {"embedding": [7]}, "r": {}}}',
        '{"chunks": {"x": {"embedding": [9]}}, "chunks": null, "version": 2}',
    ]
    for i, doc in enumerate(dup_docs):
        path = str(tmp_path / f"dupid{i}.json")
        open(path, "w").write(doc)
        rows, meta = _native_read(rlr, path, 3)
        want_state, want_rows = persistence._read_state(path, 3, native=False)
        assert json.loads(meta) == {k: ({ck: ({**cv, "embedding": []} if isinstance(cv, dict) and "embedding" in cv else cv)
                                         for ck, cv in v.items()} if k == "chunks" and isinstance(v, dict) else v)
                                    for k, v in want_state.items()}, doc
        assert rows.shape == want_rows.shape and np.array_equal(bits(rows), bits(want_rows)), doc
    # malformed input is an error, not a guess: numbers follow the JSON grammar (serde_json and the json module refuse
    # inf / nan / 01 / 1. / .5 / +1, doubled and trailing commas, nested arrays)
    for lit in ("inf", "-inf", "Infinity", "nan", "NaN", "01", "1.", ".5", "+1", "1e", "1e+", "0x10", "--1", "[1]", '"1"', "1,,2", "1,"):
        path = str(tmp_path / "badnum.json")
        open(path, "w").write('{"chunks": {"k": {"embedding": [%s]}}}' % lit)
        with pytest.raises(rlr.RlrError):
            _native_read(rlr, path, 3)
    # literals beyond binary64: +-inf or +-0 by magnitude, with or without an exponent (as serde_json's f64 path)
    path = str(tmp_path / "huge.json")
    open(path, "w").write('{"chunks": {"k": {"embedding": [%s, -%s, 1e-400, -0.%s1, 1e400, 0.%s1e500]}}}'
                          % ("9" * 400, "9" * 400, "0" * 400, "0" * 20))
    rows, _ = _native_read(rlr, path, 6)   # (the json module cannot be the comparator here: a 400-digit integer is a Python int)
    assert rows[0, 0] == np.inf and rows[0, 1] == -np.inf and rows[0, 2] == 0 and bits(rows)[0, 3] == 0x80000000 and rows[0, 4] == np.inf and rows[0, 5] == np.inf
    for bad in ('{"chunks": {"k": {"embedding": [1, 2', '{"chunks": {"k": {"embedding": [1, x]}}}', "[1, 2]", ""):
        path = str(tmp_path / "bad.json")
        open(path, "w").write(bad)
        with pytest.raises(rlr.RlrError):
            _native_read(rlr, path, 3)


def test_embedding_text_is_the_shortest_round_trip_decimal(rlr):
    import ctypes as C
    N = rlr._native
    rng = np.random.default_rng(12)
    vals = np.concatenate([rng.standard_normal(500).astype(np.float32),
                           rng.integers(0, 2 ** 32, 500, dtype=np.uint64).astype(np.uint32).view(np.float32),
                           np.array([0.0, -0.0, 1.0, -3.0, 16777216.0, 1e-45, 3.4028235e38, 0.1, 1 / 3], np.float32)])
    finite = np.isfinite(vals)
    buf = C.create_string_buffer(vals.size * 40 + 64)
    n = N.lib().rlr_json_format_embedding(vals.ctypes.data_as(N.f32p), vals.size, 6, buf, len(buf))
    text = buf.raw[:n].decode("ascii")
    lines = text.split("\n")
    assert lines[0] == "[" and lines[-1] == "      ]" and len(lines) == vals.size + 2
    assert all(l.startswith("        ") and not l.startswith("         ") for l in lines[1:-1])
    back = json.loads(text)
    got = np.array([np.nan if x is None else x for x in back], dtype=np.float64).astype(np.float32)
    assert np.array_equal(bits(got[finite]), bits(vals[finite]))              # reads back as the same binary32
    assert all(back[i] is None for i in np.flatnonzero(~finite))               # serde_json writes null for NaN / inf
    for i in np.flatnonzero(finite)[:200]:                                     # and no shorter decimal does
        s = lines[1 + i].strip().rstrip(",")
        mant = s.split("e")[0].replace("-", "").replace(".", "").lstrip("0")
        assert len(mant.rstrip("0")) <= 9
        assert "." in s or "e" in s                                           # floats stay floats ("1.0", not "1")
    assert N.lib().rlr_json_format_embedding(vals.ctypes.data_as(N.f32p), 0, 4, buf, len(buf)) == 2 and buf.raw[:2] == b"[]"
    assert N.lib().rlr_json_format_embedding(vals.ctypes.data_as(N.f32p), 10, 4, buf, 5) > 5   # too small: size only


def test_streaming_reader_is_faster_than_whole_file_parsing(rlr, tmp_path):
    """4000 chunks x 768 (a 40 MB file): same rows as the json module, in a fraction of its time"""
    import time
    persistence = __import__("importlib").import_module("rust-local-rag_amd.persistence")
    n, dim = 4000, 768
    raw = np.random.default_rng(1).standard_normal((n, dim)).astype(np.float32)
    path = str(tmp_path / "big.json")
    with open(path, "w") as f:
        f.write('{"version": 2, "model": "m", "chunks": {')
        for i in range(n):
            f.write(("," if i else "") + json.dumps(f"id{i}") + ": " +
                    json.dumps({"id": f"id{i}", "text": f"t{i}", "embedding": [float(x) for x in raw[i]], "chunk_index": i}))
        f.write('}, "needs_reindex": false}')
    t0 = time.perf_counter()
    rows, meta = _native_read(rlr, path, dim)
    t_native = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, want = persistence._read_state(path, dim, native=False)
    t_python = time.perf_counter() - t0
    assert np.array_equal(bits(rows), bits(want)) and np.array_equal(bits(rows), bits(raw))
    assert len(json.loads(meta)["chunks"]) == n
    assert t_native < t_python, (t_native, t_python)


@pytest.mark.gpu
def test_metadata_survives_load_save_load_and_both_readers_agree(rlr, oracle, tmp_path):
    """ChunkMetadata written back unchanged (rag_engine.rs:1477-1518, :1699-1706), also through the legacy migration;
    the native and the json-module readers give byte-identical device rows."""
    dim, n, model = 64, 40, "meta-model"
    raw = (np.random.default_rng(8).standard_normal((n, dim)) * 2).astype(np.float32)
    chunks = {}
    for i in range(n):
        c = _chunk(f"k{i}", f"doc{i % 3}.pdf", raw[i], i, page=2 + i)
        c["metadata"] = {"page_range": [2 + i, 3 + i], "sentence_range": [i, i + 4], "section_title": f"S {i}" if i % 2 else None,
                         "token_count": 100 + i, "overlap_with_previous": i % 3}
        c["section"] = f"sec{i}" if i % 5 == 0 else None
        chunks[f"k{i}"] = c
    hashes = {"doc0.pdf": "a", "doc1.pdf": "b", "doc2.pdf": "c"}
    _write(rlr.get_legacy_path(str(tmp_path)), {"version": 2, "model": model, "chunks": chunks, "needs_reindex": False,
                                                "document_hashes": hashes})
    a = rlr.RagEngine(dim)
    rep = rlr.load_from_disk(a, str(tmp_path), model)                       # legacy -> migrated model file
    assert rep.migrated and rep.n_chunks == n
    migrated = json.load(open(rlr.get_index_path(str(tmp_path), model)))
    assert list(migrated["chunks"].keys()) == list(chunks.keys())
    for cid, c in chunks.items():
        m = migrated["chunks"][cid]
        assert m["metadata"] == c["metadata"] and m["section"] == c["section"] and m["page_number"] == c["page_number"]
        assert m["text"] == c["text"] and m["chunk_index"] == c["chunk_index"]
    assert migrated["document_hashes"] == hashes and migrated["model"] == model and migrated["version"] == 2
    b, c_ = rlr.RagEngine(dim), rlr.RagEngine(dim)
    rb = rlr.load_from_disk(b, str(tmp_path), model, native=True)
    rc = rlr.load_from_disk(c_, str(tmp_path), model, native=False)
    assert rb.n_chunks == rc.n_chunks == n and rb.document_hashes == rc.document_hashes
    assert np.array_equal(bits(b.index.fetch_rows(np.arange(n))), bits(c_.index.fetch_rows(np.arange(n))))
    assert [ch.metadata for ch in b._chunks] == [c["metadata"] for c in chunks.values()]
    rlr.save_to_disk(b, str(tmp_path), model, document_hashes=rb.document_hashes)
    again = json.load(open(rlr.get_index_path(str(tmp_path), model)))
    assert {k: v["metadata"] for k, v in again["chunks"].items()} == {k: v["metadata"] for k, v in chunks.items()}
    # the written file is what json.dumps(indent=2) would lay out, with one embedding value per line
    text = open(rlr.get_index_path(str(tmp_path), model)).read()
    assert text.startswith('{\n  "version": 2,\n  "model": "meta-model",\n  "chunks": {\n    "k0": {\n      "id": "k0",')
    assert '\n      "embedding": [\n        ' in text and text.endswith("\n}")
    for e in (a, b, c_):
        e.close()
