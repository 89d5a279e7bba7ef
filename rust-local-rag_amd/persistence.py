"""Corpus loader / saver: `chunks_{model}.json`  <->  dense matrix in HBM + id table.

SURVEY.md section 8(f) row f1 -- the step *before* the hot path.  Mirrors the reference's
model-partitioned persistence (src/rag_engine.rs:1427-1709):

  sanitize_model_name   :1435-1461      get_index_path :1465-1468     get_legacy_path :1471-1473
  save_to_disk          :1477-1518      (version 2, pretty JSON, atomic tmp + rename)
  load_from_disk        :1520-1653      (model-specific file first; legacy chunks.json only when its
                                         model matches; never delete another model's data)
  apply_loaded_state    :1655-1709      (version < 2 -> clear + needs_reindex; re-normalise EVERY
                                         embedding on load, :1678-1680)

The re-normalise-on-load quirk is reproduced on the GPU: rows are uploaded raw and
`rlr_index_upload(normalize_on_device=1)` applies the reference `normalize` bit-identically, so
the device rows equal what the reference would hold in memory after the same load.

Row order: the reference keeps chunks in a HashMap (arbitrary order); here row = position of the
chunk in the file's `chunks` object, which defines the tie order of searches ("lower row first").
JSON numbers go through binary64 and are rounded to binary32, like serde_json's f32 path.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

from . import _native as N
from .engine import DocumentChunk, RagEngine


def sanitize_model_name(model_name: str) -> str:
    """rag_engine.rs:1435-1461"""
    trimmed = model_name.strip()
    if not trimmed:
        return "default"
    sanitized = "".join(c if (c.isascii() and (c.isalnum() or c in "-_.")) else "_" for c in trimmed)
    if not sanitized or all(c in "_." for c in sanitized):
        return "default"
    return sanitized


def get_index_path(data_dir: str, model_name: str) -> str:
    """rag_engine.rs:1465-1468"""
    return os.path.join(data_dir, f"chunks_{sanitize_model_name(model_name)}.json")


def get_legacy_path(data_dir: str) -> str:
    """rag_engine.rs:1471-1473"""
    return os.path.join(data_dir, "chunks.json")


@dataclass
class LoadReport:
    source: Optional[str] = None     # file the state came from (None: started fresh)
    n_chunks: int = 0
    needs_reindex: bool = False
    migrated: bool = False           # legacy chunks.json re-saved in the model-specific format
    document_hashes: Dict[str, str] = field(default_factory=dict)


DEFAULT_METADATA = {"page_range": None, "sentence_range": None, "section_title": None, "token_count": 0,
                    "overlap_with_previous": 0}


def _embedding_text(row: np.ndarray, indent: int, buf) -> str:
    """one embedding array as serde_json's pretty printer lays it out, every value the shortest decimal that reads
    back as the same binary32 (rlr_json_format_embedding: std::to_chars; non-finite -> null)"""
    row = np.ascontiguousarray(row, dtype=np.float32)
    L = N.lib()
    n = L.rlr_json_format_embedding(row.ctypes.data_as(N.f32p), row.size, indent, buf, len(buf))
    if n > len(buf):
        raise RuntimeError("embedding text buffer too small")
    return buf.raw[:n].decode("ascii")


def _member(key: str, value, indent: int) -> str:
    """`"key": value` with nested containers indented like json.dumps(indent=2) at depth `indent`"""
    text = json.dumps(value, indent=2, ensure_ascii=False)
    if "\n" in text:
        text = text.replace("\n", "\n" + " " * indent)
    return " " * indent + json.dumps(key, ensure_ascii=False) + ": " + text


def save_to_disk(engine: RagEngine, data_dir: str, model_name: str, needs_reindex: bool = False,
                 document_hashes: Optional[Dict[str, str]] = None, metadata: Optional[Dict[str, dict]] = None) -> str:
    """rag_engine.rs:1477-1518: version 2 state, pretty-printed, written to `<final>.json.tmp` then renamed.
    One streaming pass: chunk objects are written one after the other with their embedding lines (rows fetched from
    the GPU in blocks), so the cost is linear in the file size.  Each chunk's `metadata` is what was loaded with it
    (DocumentChunk.metadata), or the entry of the `metadata` argument for its id, or the defaults."""
    final_path = get_index_path(data_dir, model_name)
    temp_path = final_path[: -len(".json")] + ".json.tmp"
    n = len(engine)
    dim = engine.dim
    buf = C.create_string_buffer(max(dim, 1) * 40 + 64)
    block = max(1, min(n, (64 << 20) // (4 * max(dim, 1))))
    with open(temp_path, "w", encoding="utf-8") as f:
        f.write("{\n")
        f.write(_member("version", 2, 2) + ",\n")
        f.write(_member("model", model_name, 2) + ",\n")
        f.write('  "chunks": {' + ("\n" if n else ""))
        for r0 in range(0, n, block):
            r1 = min(n, r0 + block)
            rows = engine.index.fetch_rows(np.arange(r0, r1, dtype=np.uint64))
            for r in range(r0, r1):
                ch = engine._chunks[r]
                md = (metadata or {}).get(ch.id) or ch.metadata or DEFAULT_METADATA
                f.write("    " + json.dumps(ch.id, ensure_ascii=False) + ": {\n")
                f.write(_member("id", ch.id, 6) + ",\n")
                f.write(_member("document_name", ch.document_name, 6) + ",\n")
                f.write(_member("text", ch.text, 6) + ",\n")
                f.write('      "embedding": ' + _embedding_text(rows[r - r0], 6, buf) + ",\n")
                f.write(_member("chunk_index", ch.chunk_index, 6) + ",\n")
                f.write(_member("page_number", ch.page_number, 6) + ",\n")
                f.write(_member("section", ch.section, 6) + ",\n")
                f.write(_member("metadata", md, 6) + "\n")
                f.write("    }" + (",\n" if r + 1 < n else "\n"))
        f.write(("  }" if n else "}") + ",\n")
        f.write(_member("needs_reindex", bool(needs_reindex), 2))
        if document_hashes:  # skip_serializing_if = "HashMap::is_empty"
            f.write(",\n" + _member("document_hashes", dict(document_hashes), 2))
        f.write("\n}")
    os.replace(temp_path, final_path)  # atomic commit
    return final_path


def _read_state(path: str, dim: int, native: bool):
    """-> (state dict whose chunks carry no embeddings, rows f32 [n, dim]) in file order.
    native: one streaming pass in C++ over the memory-mapped file (rlr_json_load_corpus); else the json module
    (the comparator the parity test holds the native reader against)."""
    if native:
        c = N.JsonCorpusC()
        N.check(N.lib().rlr_json_load_corpus(os.fsencode(path), dim, C.byref(c)))
        try:
            n = int(c.n_rows)
            rows = np.ctypeslib.as_array(c.rows, shape=(n * dim,)).reshape(n, dim).copy() if n else \
                np.zeros((0, dim), np.float32)
            state = json.loads(C.string_at(c.meta_json, c.meta_len).decode("utf-8"))
        finally:
            N.lib().rlr_json_free_corpus(C.byref(c))
        chunks = state.get("chunks", {}) if isinstance(state, dict) else {}
        if rows.shape[0] != (len(chunks) if isinstance(chunks, dict) else 0):
            # the two readers disagree about what the file's chunk map holds: never pair rows with the wrong chunks
            raise N.RlrError(N.RLR_E_INVALID, f"{path}: {rows.shape[0]} embedding rows for "
                                              f"{len(chunks) if isinstance(chunks, dict) else 0} chunks")
        return state, rows
    with open(path, encoding="utf-8") as f:
        state = json.load(f)
    chunks = state.get("chunks", {}) if isinstance(state, dict) else {}
    rows = np.zeros((len(chunks) if isinstance(chunks, dict) else 0, dim), dtype=np.float32)
    if isinstance(chunks, dict):
        for r, c in enumerate(chunks.values()):
            if not isinstance(c, dict):
                continue
            emb = np.asarray(c.get("embedding", []), dtype=np.float64).astype(np.float32)
            m = min(emb.size, dim)  # dot_product's zip truncates / a short row contributes zeros
            rows[r, :m] = emb[:m]
    return state, rows


def _apply_loaded_state(engine: RagEngine, state: dict, rows: np.ndarray, source: str, data_dir: str, model_name: str,
                        migrate: bool) -> LoadReport:
    """rag_engine.rs:1655-1709"""
    rep = LoadReport(source=source)
    version = int(state.get("version", 0))
    if version < 2:
        # outdated format: clear and mark for reindex (:1664-1672)
        engine.index.upload(np.zeros((0, engine.dim), np.float32))
        engine.lexical.clear()
        engine._chunks, engine._row_of = [], {}
        rep.needs_reindex = True
        save_to_disk(engine, data_dir, model_name, needs_reindex=True)
        return rep
    chunks = state.get("chunks", {})
    ids = list(chunks.keys())
    n = len(ids)
    metas = []
    for r, cid in enumerate(ids):
        c = chunks[cid]
        md = c.get("metadata")
        metas.append(DocumentChunk(c.get("id", cid), c.get("document_name", ""), c.get("text", ""),
                                   int(c.get("chunk_index", 0)), int(c.get("page_number", 0)), c.get("section"),
                                   md if isinstance(md, dict) else None))
    # `for chunk in self.chunks.values_mut() { normalize(&mut chunk.embedding) }` (:1678-1680), on the GPU
    engine.index.upload(rows, normalize=True)
    engine._chunks = metas
    engine._row_of = {ch.id: r for r, ch in enumerate(metas)}
    # validate_index_sync (:1378-1388): every chunk is (re-)added to the lexical index
    engine.lexical.clear()
    for r, ch in enumerate(metas):
        engine.lexical.add_chunk(r, ch.text)
    rep.n_chunks = n
    rep.needs_reindex = bool(state.get("needs_reindex", False))
    rep.document_hashes = dict(state.get("document_hashes", {}))
    if not rep.document_hashes and n:  # :1686-1691
        rep.needs_reindex = True
    # validate_index_sync (:1373-1425): drop hashes of documents that have no chunks left
    docs = {ch.document_name for ch in metas}
    rep.document_hashes = {d: h for d, h in rep.document_hashes.items() if d in docs}
    if migrate:  # :1699-1706  legacy file is preserved; the chunks' metadata goes back out as it came in
        save_to_disk(engine, data_dir, model_name, rep.needs_reindex, rep.document_hashes)
        rep.migrated = True
    return rep


# ---- binary side-car cache ("and a binary side-car cache") ---------------------------------------
# Parsing the pretty-printed `Vec<f32>` of a large corpus dominates start-up.  After a successful load the rows the
# device holds (i.e. AFTER the load-time re-normalisation) can be written next to the JSON as raw binary32 plus the
# chunk metadata; the next load of the SAME file (size + mtime_ns recorded in the header) uploads them as they are
# and never touches the embedding arrays.  Any mismatch -- other size / mtime / model / dim / version -- ignores the
# cache.  The JSON stays the source of truth; the cache is disposable.
SIDECAR_MAGIC = b"RLRCACHE1\n"


def get_sidecar_path(data_dir: str, model_name: str) -> str:
    return os.path.join(data_dir, f"chunks_{sanitize_model_name(model_name)}.rlrcache")


def _file_identity(path: str) -> Dict[str, int]:
    st = os.stat(path)
    return {"size": int(st.st_size), "mtime_ns": int(st.st_mtime_ns)}


def save_sidecar(engine: RagEngine, data_dir: str, model_name: str, report: "LoadReport") -> str:
    """Write the cache for the state just loaded from / saved to `get_index_path(data_dir, model_name)`."""
    src = get_index_path(data_dir, model_name)
    n = len(engine)
    rows = engine.index.fetch_rows(np.arange(n, dtype=np.uint64)) if n else np.zeros((0, engine.dim), np.float32)
    header = {"model": model_name, "dim": engine.dim, "n": n, "source": _file_identity(src),
              "needs_reindex": bool(report.needs_reindex), "document_hashes": report.document_hashes,
              "chunks": [[ch.id, ch.document_name, ch.text, ch.chunk_index, ch.page_number, ch.section, ch.metadata]
                         for ch in engine._chunks]}
    blob = json.dumps(header, ensure_ascii=False).encode("utf-8")
    path = get_sidecar_path(data_dir, model_name)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(SIDECAR_MAGIC)
        f.write(len(blob).to_bytes(8, "little"))
        f.write(blob)
        f.write(np.ascontiguousarray(rows, dtype="<f4").tobytes())
    os.replace(tmp, path)
    return path


def _load_sidecar(engine: RagEngine, data_dir: str, model_name: str) -> Optional[LoadReport]:
    path, src = get_sidecar_path(data_dir, model_name), get_index_path(data_dir, model_name)
    if not (os.path.exists(path) and os.path.exists(src)):
        return None
    try:
        with open(path, "rb") as f:
            if f.read(len(SIDECAR_MAGIC)) != SIDECAR_MAGIC:
                return None
            hlen = int.from_bytes(f.read(8), "little")
            header = json.loads(f.read(hlen).decode("utf-8"))
            if (header.get("model") != model_name or header.get("dim") != engine.dim
                    or header.get("source") != _file_identity(src)):
                return None
            n = int(header["n"])
            rows = np.frombuffer(f.read(n * engine.dim * 4), dtype="<f4")
            if rows.size != n * engine.dim or len(header["chunks"]) != n:
                return None
    except (OSError, ValueError, KeyError):
        return None
    engine.index.upload(rows.reshape(n, engine.dim), normalize=False)   # already the post-load rows
    engine._chunks = [DocumentChunk(*c) for c in header["chunks"]]
    engine._row_of = {ch.id: r for r, ch in enumerate(engine._chunks)}
    engine.lexical.clear()
    for r, ch in enumerate(engine._chunks):
        engine.lexical.add_chunk(r, ch.text)
    return LoadReport(source=path, n_chunks=n, needs_reindex=bool(header.get("needs_reindex", False)),
                      document_hashes=dict(header.get("document_hashes", {})))


def load_from_disk(engine: RagEngine, data_dir: str, model_name: str, use_sidecar: bool = False,
                   native: bool = True) -> LoadReport:
    """rag_engine.rs:1520-1653.  use_sidecar: take the binary cache when it matches the JSON file, and (re)write it
    after a JSON load.  native: read the file with the streaming C++ reader (default) or with the json module."""
    model_path = get_index_path(data_dir, model_name)
    legacy_path = get_legacy_path(data_dir)
    if use_sidecar:
        rep = _load_sidecar(engine, data_dir, model_name)
        if rep is not None:
            return rep
        rep = load_from_disk(engine, data_dir, model_name, use_sidecar=False, native=native)
        if rep.source == model_path and os.path.exists(model_path):
            save_sidecar(engine, data_dir, model_name, rep)
        return rep
    if os.path.exists(model_path):
        try:
            state, rows = _read_state(model_path, engine.dim, native)
            if not isinstance(state, dict) or "version" not in state or "model" not in state or "chunks" not in state:
                raise ValueError("missing field")
        except (ValueError, OSError, N.RlrError):
            # corrupted model-specific file: keep it for inspection, start empty, mark for reindex (:1571-1585)
            return LoadReport(source=None, needs_reindex=True)
        return _apply_loaded_state(engine, state, rows, model_path, data_dir, model_name, migrate=False)
    if os.path.exists(legacy_path):
        try:
            state, rows = _read_state(legacy_path, engine.dim, native)
        except (ValueError, OSError, N.RlrError):
            state, rows = None, None
        if isinstance(state, dict) and isinstance(state.get("model"), str):
            if state["model"] == model_name and "version" in state and "chunks" in state:
                return _apply_loaded_state(engine, state, rows, legacy_path, data_dir, model_name, migrate=True)
            # belongs to another model: preserved, start fresh (:1620-1628)
            return LoadReport()
        if isinstance(state, dict) and state and all(isinstance(v, dict) for v in state.values()):
            # very old format: raw chunk map without a model field -> reindex required (:1630-1646)
            return LoadReport(needs_reindex=True)
    return LoadReport()
