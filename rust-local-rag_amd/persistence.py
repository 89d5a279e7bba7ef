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

import json
import os
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

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


def _f32_text(x: np.float32) -> str:
    """shortest decimal that round-trips the binary32 value (what serde_json prints for f32)"""
    return np.format_float_positional(x, unique=True, trim="0") if np.isfinite(x) else "null"


def save_to_disk(engine: RagEngine, data_dir: str, model_name: str, needs_reindex: bool = False,
                 document_hashes: Optional[Dict[str, str]] = None, metadata: Optional[Dict[str, dict]] = None) -> str:
    """rag_engine.rs:1477-1518: version 2 state, pretty-printed, written to `<final>.json.tmp` then renamed."""
    final_path = get_index_path(data_dir, model_name)
    temp_path = final_path[: -len(".json")] + ".json.tmp"
    n = len(engine)
    rows = engine.index.fetch_rows(np.arange(n, dtype=np.uint64)) if n else np.zeros((0, engine.dim), np.float32)
    chunks = {}
    for r, ch in enumerate(engine._chunks):
        md = (metadata or {}).get(ch.id) or {"page_range": None, "sentence_range": None, "section_title": None,
                                            "token_count": 0, "overlap_with_previous": 0}
        chunks[ch.id] = {"id": ch.id, "document_name": ch.document_name, "text": ch.text,
                         "embedding": f"@@EMB{r}@@", "chunk_index": ch.chunk_index,
                         "page_number": ch.page_number, "section": ch.section, "metadata": md}
    state = {"version": 2, "model": model_name, "chunks": chunks, "needs_reindex": bool(needs_reindex)}
    if document_hashes:  # skip_serializing_if = "HashMap::is_empty"
        state["document_hashes"] = dict(document_hashes)
    text = json.dumps(state, indent=2, ensure_ascii=False)
    # embeddings are spliced in as shortest-round-trip binary32 literals, one per line like
    # serde_json's pretty printer (json.dumps would print the binary64 expansion of each value)
    pad, pad_close = " " * 8, " " * 6
    for r in range(n):
        body = ",\n".join(pad + _f32_text(v) for v in rows[r])
        text = text.replace(f'"@@EMB{r}@@"', "[\n" + body + "\n" + pad_close + "]" if rows.shape[1] else "[]", 1)
    with open(temp_path, "w", encoding="utf-8") as f:
        f.write(text)
    os.replace(temp_path, final_path)  # atomic commit
    return final_path


def _apply_loaded_state(engine: RagEngine, state: dict, source: str, data_dir: str, model_name: str,
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
    rows = np.zeros((n, engine.dim), dtype=np.float32)
    metas = []
    for r, cid in enumerate(ids):
        c = chunks[cid]
        emb = np.asarray(c.get("embedding", []), dtype=np.float64).astype(np.float32)
        m = min(emb.size, engine.dim)  # dot_product's zip truncates / a short row contributes zeros
        rows[r, :m] = emb[:m]
        metas.append(DocumentChunk(c.get("id", cid), c.get("document_name", ""), c.get("text", ""),
                                   int(c.get("chunk_index", 0)), int(c.get("page_number", 0)), c.get("section")))
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
    if migrate:  # :1699-1706  legacy file is preserved
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
              "chunks": [[ch.id, ch.document_name, ch.text, ch.chunk_index, ch.page_number, ch.section]
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


def load_from_disk(engine: RagEngine, data_dir: str, model_name: str, use_sidecar: bool = False) -> LoadReport:
    """rag_engine.rs:1520-1653.  use_sidecar: take the binary cache when it matches the JSON file, and (re)write it
    after a JSON load."""
    model_path = get_index_path(data_dir, model_name)
    legacy_path = get_legacy_path(data_dir)
    if use_sidecar:
        rep = _load_sidecar(engine, data_dir, model_name)
        if rep is not None:
            return rep
        rep = load_from_disk(engine, data_dir, model_name, use_sidecar=False)
        if rep.source == model_path and os.path.exists(model_path):
            save_sidecar(engine, data_dir, model_name, rep)
        return rep
    if os.path.exists(model_path):
        try:
            with open(model_path, encoding="utf-8") as f:
                state = json.load(f)
            if not isinstance(state, dict) or "version" not in state or "model" not in state or "chunks" not in state:
                raise ValueError("missing field")
        except (ValueError, OSError):
            # corrupted model-specific file: keep it for inspection, start empty, mark for reindex (:1571-1585)
            return LoadReport(source=None, needs_reindex=True)
        return _apply_loaded_state(engine, state, model_path, data_dir, model_name, migrate=False)
    if os.path.exists(legacy_path):
        try:
            with open(legacy_path, encoding="utf-8") as f:
                state = json.load(f)
        except (ValueError, OSError):
            state = None
        if isinstance(state, dict) and isinstance(state.get("model"), str):
            if state["model"] == model_name and "version" in state and "chunks" in state:
                return _apply_loaded_state(engine, state, legacy_path, data_dir, model_name, migrate=True)
            # belongs to another model: preserved, start fresh (:1620-1628)
            return LoadReport()
        if isinstance(state, dict) and state and all(isinstance(v, dict) for v in state.values()):
            # very old format: raw chunk map without a model field -> reindex required (:1630-1646)
            return LoadReport(needs_reindex=True)
    return LoadReport()
