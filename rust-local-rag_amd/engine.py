"""RagEngine: the reference's search-side operator interface over the GPU index.

Mirrors rust-local-rag `RagEngine::{search, search_with_diversity,
get_embedding_candidates}` (src/rag_engine.rs:470-701, :717-759, :415-461) and the API
layer `search_documents` / `http_search` / `format_search_results`
(src/mcp_server.rs:81-110, :371-389, :599-637) -- same names, argument meaning and
edge-case behaviour.  The numerics run in librlr_gpu.so (csrc/engine.cpp + HIP kernels);
this file only carries chunk metadata and marshals arguments.

Outside this path and therefore passed IN by the caller: the query *embedding* (the
reference obtains it from Ollama, embeddings.rs:91-102), BM25 candidates
(`LexicalIndex::score`) as (chunk_id, score) pairs, and the LLM reranker.
"""
from __future__ import annotations

import ctypes as C
import uuid
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _native as N
from .index import GpuIndex, _f32
from .lexical import LexicalIndex, tokenize


@dataclass
class QueryWeights:  # rag_engine.rs:1846-1865
    embedding: Optional[float] = None
    lexical: Optional[float] = None
    reranker: Optional[float] = None
    initial: Optional[float] = None

    def to_c(self) -> N.QueryWeightsC:
        c = N.QueryWeightsC()
        for name in ("embedding", "lexical", "reranker", "initial"):
            v = getattr(self, name)
            setattr(c, "has_" + name, int(v is not None))
            setattr(c, name, float(v) if v is not None else 0.0)
        return c


@dataclass
class ResolvedWeights:  # rag_engine.rs:1877-1896
    embedding: float
    lexical: float
    reranker: float
    initial: float

    @staticmethod
    def from_query_weights(weights: Optional[QueryWeights]) -> "ResolvedWeights":
        out = N.ResolvedWeightsC()
        if weights is None:
            N.lib().rlr_resolve_weights(None, C.byref(out))
        else:
            c = weights.to_c()
            N.lib().rlr_resolve_weights(C.byref(c), C.byref(out))
        return ResolvedWeights(out.embedding, out.lexical, out.reranker, out.initial)


def resolve_weight(override: Optional[float], default: float) -> float:
    """rag_engine.rs:1869-1873"""
    has = override is not None
    return float(N.lib().rlr_resolve_weight(int(has), float(override) if has else 0.0, float(default)))


def normalize(v) -> np.ndarray:
    """rag_engine.rs:1763-1771 (host side, reference order)"""
    a = _f32(v).copy()
    N.lib().rlr_normalize(a.ctypes.data_as(N.f32p), a.size)
    return a


@dataclass
class DocumentChunk:  # rag_engine.rs:46-59 (embedding lives in HBM, not here)
    id: str
    document_name: str
    text: str
    chunk_index: int
    page_number: int = 0
    section: Optional[str] = None
    metadata: Optional[dict] = None  # ChunkMetadata (:61-70) as loaded; written back unchanged by save_to_disk


@dataclass
class SearchResult:  # rag_engine.rs:72-100
    text: str
    score: float
    document: str
    chunk_id: str
    chunk_index: int
    page_number: int
    section: Optional[str] = None
    embedding_score: Optional[float] = None
    lexical_score: Optional[float] = None
    initial_score: Optional[float] = None
    reranker_score: Optional[float] = None
    yes_logprob: Optional[float] = None
    no_logprob: Optional[float] = None
    row: int = -1  # position in the GPU matrix (not part of the reference struct)

    def to_json(self) -> dict:
        """serde shape incl. skip_serializing_if = Option::is_none (rag_engine.rs:83-99)"""
        d = {"text": self.text, "score": self.score, "document": self.document, "chunk_id": self.chunk_id,
             "chunk_index": self.chunk_index, "page_number": self.page_number, "section": self.section}
        for k in ("embedding_score", "lexical_score", "initial_score", "reranker_score", "yes_logprob", "no_logprob"):
            v = getattr(self, k)
            if v is not None:
                d[k] = v
        return d


@dataclass
class SearchRequest:  # mcp_server.rs:17-31
    query_embedding: Sequence[float]
    top_k: Optional[int] = None
    diversity_factor: Optional[float] = None
    weights: Optional[QueryWeights] = None
    lexical: Sequence[Tuple[str, float]] = field(default_factory=list)
    query: Optional[str] = None   # the query text (mcp_server.rs:19); scored by the GPU LexicalIndex when given


class RagEngine:
    """Search half of the reference's RagEngine with the embeddings resident on one GPU."""

    def __init__(self, dim: int, dtype: str = "f32", device: int = 0, accelerate: Optional[str] = None):
        """accelerate: None | "image" | "q8" -- keep an optional nomination copy of the rows so the scan streams
        a half / a quarter of the bytes (GpuIndex.enable_batch_image); results are unchanged"""
        self.index = GpuIndex(dim, dtype, device)
        if accelerate == "image":
            self.index.enable_batch_image(True, single_query=True)
        elif accelerate == "q8":
            self.index.enable_batch_image(False, q8=True)
        elif accelerate is not None:
            raise ValueError("accelerate must be None, 'image' or 'q8'")
        self.lexical = LexicalIndex(device)          # BM25 postings in HBM (rag_engine.rs:112)
        self.dim = dim
        self._chunks: List[DocumentChunk] = []       # row -> chunk
        self._row_of: Dict[str, int] = {}            # chunk_id -> row

    def close(self) -> None:
        self.index.close()
        self.lexical.close()

    # -- index mutation (sites rag_engine.rs:347-348, :358-384) --------------------------
    def add_document(self, document_name: str, texts: Sequence[str], embeddings, pages: Optional[Sequence[int]] = None,
                     sections: Optional[Sequence[Optional[str]]] = None) -> List[str]:
        """Replace `document_name`'s chunks: drop its old rows, normalise and append the new ones."""
        self.remove_document(document_name)
        emb = _f32(embeddings).reshape(len(texts), self.dim)
        first = self.index.append(emb, normalize=True)  # normalize(&mut embedding) :358-359, on the GPU
        ids = []
        for i, text in enumerate(texts):
            cid = str(uuid.uuid4())
            ch = DocumentChunk(cid, document_name, text, i, pages[i] if pages else 0, sections[i] if sections else None)
            self._chunks.append(ch)
            self._row_of[cid] = first + i
            self.lexical.add_chunk(first + i, text)   # lexical_index.add_chunk(&chunk.id, &chunk.text) :382
            ids.append(cid)
        return ids

    def remove_document(self, document_name: str) -> int:
        dead = [r for r, ch in enumerate(self._chunks) if ch.document_name == document_name]
        if dead:
            self.index.delete_rows(dead)  # chunks.retain(|_, c| c.document_name != filename)
            self.lexical.remove_rows(dead)
            dead_set = set(dead)
            self._chunks = [ch for r, ch in enumerate(self._chunks) if r not in dead_set]
            self._row_of = {ch.id: r for r, ch in enumerate(self._chunks)}
        return len(dead)

    def __len__(self) -> int:
        return len(self._chunks)

    # -- helpers ---------------------------------------------------------------------------
    def _lex(self, lexical: Sequence[Tuple[str, float]], query_text: Optional[str] = None, limit: int = 0):
        if query_text is not None:
            # `self.lexical_index.score(query, top_k.saturating_mul(5))` :505, on the GPU
            lr, ls = self.lexical.score(query_text, limit)
            if lr.size == 0:
                return np.zeros(1, np.uint64), np.zeros(1, np.float32), 0
            return np.ascontiguousarray(lr), np.ascontiguousarray(ls), int(lr.size)
        rows = [self._row_of[cid] for cid, _ in lexical if cid in self._row_of]
        scores = [s for cid, s in lexical if cid in self._row_of]
        lr = np.ascontiguousarray(rows if rows else [0], dtype=np.uint64)
        ls = np.ascontiguousarray(scores if scores else [0], dtype=np.float32)
        return lr, ls, len(rows)

    _HIT_DTYPE = np.dtype([("row", "<u8"), ("score", "<f4"), ("embedding_score", "<f4"), ("lexical_score", "<f4"),
                           ("initial_score", "<f4")])   # = _native.SearchHitC

    def _results(self, hits, n: int) -> List[SearchResult]:
        if n == 0:
            return []
        # one view over the ctypes array and one tolist() instead of 5 n attribute reads (a third of a 0.3 ms search)
        rows = np.frombuffer(hits, dtype=self._HIT_DTYPE, count=n).tolist()
        chunks = self._chunks
        out = []
        for row, score, emb, lex, init in rows:
            ch = chunks[row]
            out.append(SearchResult(ch.text, score, ch.document_name, ch.id, ch.chunk_index, ch.page_number, ch.section,
                                    emb, lex, init, None, None, None, row))
        return out

    # -- RagEngine::search (rag_engine.rs:470-701) -----------------------------------------
    def search(self, query_embedding, top_k: int, weights: Optional[QueryWeights] = None,
               lexical: Sequence[Tuple[str, float]] = (), stage: int = 0,
               query_text: Optional[str] = None) -> List[SearchResult]:
        """`lexical`: BM25 pairs computed by the caller, or `query_text`: scored by the GPU LexicalIndex."""
        q = _f32(query_embedding).ravel()
        cap = max(3 * max(top_k, 1), 1)
        hits = (N.SearchHitC * cap)()
        n = C.c_uint32()
        wc = weights.to_c() if weights is not None else None
        if query_text is not None:
            # BM25 of the text beside the scan, blended on the device: one enqueue, one synchronisation
            tok = " ".join(tokenize(query_text)).encode("utf-8")
            N.check(N.lib().rlr_engine_search_text(self.index.handle, self.lexical._h, q.ctypes.data_as(N.f32p), q.size, tok,
                                                   len(tok), top_k, 0.0, stage, C.byref(wc) if wc is not None else None, hits,
                                                   cap, C.byref(n)))
            return self._results(hits, n.value)
        lr, ls, nl = self._lex(lexical, None, 5 * max(top_k, 1))   # top_k.max(1) then saturating_mul(5) :490, :505
        N.check(N.lib().rlr_engine_search(self.index.handle, q.ctypes.data_as(N.f32p), q.size, top_k,
                                          C.byref(wc) if wc is not None else None, lr.ctypes.data_as(N.u64p),
                                          ls.ctypes.data_as(N.f32p), nl, stage, hits, cap, C.byref(n)))
        return self._results(hits, n.value)

    # -- RagEngine::search_with_diversity (rag_engine.rs:717-759) --------------------------
    def search_with_diversity(self, query_embedding, top_k: int, diversity_factor: float,
                              weights: Optional[QueryWeights] = None,
                              lexical: Sequence[Tuple[str, float]] = (),
                              query_text: Optional[str] = None) -> List[SearchResult]:
        q = _f32(query_embedding).ravel()
        cap = max(3 * max(top_k, 1), top_k + 10)
        hits = (N.SearchHitC * cap)()
        n = C.c_uint32()
        lam = min(max(float(diversity_factor), 0.0), 1.0)
        k_eff = top_k if lam == 0.0 else max(3 * top_k, top_k + 10)  # the top_k `search` sees (:728-735)
        wc = weights.to_c() if weights is not None else None
        if query_text is not None:
            tok = " ".join(tokenize(query_text)).encode("utf-8")
            N.check(N.lib().rlr_engine_search_text(self.index.handle, self.lexical._h, q.ctypes.data_as(N.f32p), q.size, tok,
                                                   len(tok), top_k, float(diversity_factor), 0,
                                                   C.byref(wc) if wc is not None else None, hits, cap, C.byref(n)))
            return self._results(hits, n.value)
        lr, ls, nl = self._lex(lexical, None, 5 * max(k_eff, 1))    # search() treats 0 as 1 (:490) before * 5 (:505)
        N.check(N.lib().rlr_engine_search_with_diversity(
            self.index.handle, q.ctypes.data_as(N.f32p), q.size, top_k, float(diversity_factor),
            C.byref(wc) if wc is not None else None, lr.ctypes.data_as(N.u64p), ls.ctypes.data_as(N.f32p), nl,
            hits, cap, C.byref(n)))
        return self._results(hits, n.value)

    # -- tail of RagEngine::search when a reranker answered (rag_engine.rs:599-700) --------------
    def finish_with_reranker(self, candidates: Sequence[SearchResult], reranked: Sequence[Tuple[str, float]],
                             top_k: int, weights: Optional[QueryWeights] = None) -> List[SearchResult]:
        """candidates: `search(..., stage=1)`; reranked: the reranker's (chunk_id, relevance) list in its
        order (empty = reranker absent/failed -> fallback ordering by initial score)."""
        n = len(candidates)
        cand = (N.SearchHitC * max(n, 1))()
        for i, r in enumerate(candidates):
            cand[i].row, cand[i].score, cand[i].embedding_score = r.row, r.score, r.embedding_score or 0.0
            cand[i].lexical_score, cand[i].initial_score = r.lexical_score or 0.0, r.initial_score or 0.0
        known = [(self._row_of[c], s) for c, s in reranked if c in self._row_of]
        rr = np.ascontiguousarray([r for r, _ in known] or [0], dtype=np.uint64)
        rs = np.ascontiguousarray([s for _, s in known] or [0], dtype=np.float32)
        top_k = max(top_k, 1)
        out = (N.SearchHitC * max(n, 1))()
        rer = np.zeros(max(n, 1), np.float32)
        has = np.zeros(max(n, 1), np.int32)
        n_out = C.c_uint32()
        wc = weights.to_c() if weights is not None else None
        N.check(N.lib().rlr_engine_blend_reranked(cand, n, rr.ctypes.data_as(N.u64p), rs.ctypes.data_as(N.f32p), len(known),
                                                  top_k, C.byref(wc) if wc is not None else None, out,
                                                  rer.ctypes.data_as(N.f32p), has.ctypes.data_as(N.i32p), max(n, 1),
                                                  C.byref(n_out)))
        res = self._results(out, n_out.value)
        for i, r in enumerate(res):
            r.reranker_score = float(rer[i]) if has[i] else None
        return res

    # -- additive batched entry point (oracle: loop search_with_diversity over the batch) -------
    def search_with_diversity_batch(self, query_embeddings, top_k: int, diversity_factor: float,
                                    weights: Optional[QueryWeights] = None) -> List[List[SearchResult]]:
        q = _f32(query_embeddings)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        nq, dq = q.shape
        cap = max(3 * max(top_k, 1), top_k + 10)
        hits = (N.SearchHitC * (cap * max(nq, 1)))()
        n_out = np.zeros(max(nq, 1), dtype=np.uint32)
        wc = weights.to_c() if weights is not None else None
        N.check(N.lib().rlr_engine_search_with_diversity_batch(
            self.index.handle, q.ctypes.data_as(N.f32p), dq, nq, top_k, float(diversity_factor),
            C.byref(wc) if wc is not None else None, hits, cap, n_out.ctypes.data_as(N.u32p)))
        out = []
        for i in range(nq):
            view = (N.SearchHitC * cap).from_buffer(hits, i * cap * C.sizeof(N.SearchHitC))
            out.append(self._results(view, int(n_out[i])))
        return out

    # -- RagEngine::get_embedding_candidates (rag_engine.rs:415-461) -----------------------
    def get_embedding_candidates(self, query_embedding, count: int) -> List[Tuple[str, float]]:
        q = _f32(query_embedding).ravel()
        rows = np.zeros(max(count, 1), dtype=np.uint64)
        sc = np.zeros(max(count, 1), dtype=np.float32)
        n = C.c_uint32()
        N.check(N.lib().rlr_engine_embedding_candidates(self.index.handle, q.ctypes.data_as(N.f32p), q.size, count,
                                                        rows.ctypes.data_as(N.u64p), sc.ctypes.data_as(N.f32p),
                                                        C.byref(n)))
        return [(self._chunks[int(rows[i])].id, float(sc[i])) for i in range(n.value)]

    # -- API layer: search_documents / http_search (mcp_server.rs:81-110, :371-389) --------
    def search_documents(self, request: SearchRequest) -> List[SearchResult]:
        top_k = min(request.top_k if request.top_k is not None else N.DEFAULT_TOP_K, N.MAX_TOP_K)
        div = request.diversity_factor if request.diversity_factor is not None else N.DEFAULT_DIVERSITY
        div = min(max(div, 0.0), 1.0)
        return self.search_with_diversity(request.query_embedding, top_k, div, request.weights, request.lexical,
                                          request.query)


def format_search_results(results: Sequence[SearchResult]) -> str:
    """mcp_server.rs:599-637"""
    if not results:
        return "No results found."
    parts = []
    for i, r in enumerate(results):
        provenance = f"{r.document} (page {r.page_number})" if r.page_number > 0 else r.document
        section = f"*Section: {r.section}*\n" if r.section is not None else ""
        x = float(np.float32(r.score) * np.float32(100.0))
        # f32::round: half away from zero
        percentage = int(np.floor(abs(x) + 0.5) * (1 if x >= 0 else -1))
        parts.append(f"**{i + 1}. [{percentage}%] {provenance}**\n{section}\n{r.text}\n")
    return "\n---\n\n".join(parts)
