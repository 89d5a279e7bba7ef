"""Eval-harness counterpart: the `/search` wire contract on top of the
GPU path, and the retrieval metrics the reference's harness computes from it.

What is mirrored (behaviour, not code):
  wire contract   POST /search  {"query": str, "top_k"?: int = 5, "diversity_factor"?: float = 0.3}
                  -> 200 {"results": [SearchResult]}        src/mcp_server.rs:345-389
                  top_k capped at MAX_TOP_K = 100 (:364, :375), diversity clamped to [0, 1] (:376),
                  default weights (:378), engine error -> 500 (:384-387); a body without "query" is
                  rejected before the handler runs (axum's Json extractor: 422, malformed JSON: 400)
  result shape    rag_engine.rs:72-100 (Option fields omitted when None)        -> SearchResult.to_json
  client parsing  eval/rag_client.py:82-105 reads chunk_id, document, page_number, text, score, section
  matching keys   eval/rag_client.py:249-262, gold keys with page tolerance eval/eval_runner.py:167-177,
                  relevance of a result = best matching gold reference :179-193
  metrics         eval/metrics.py:18-120 -- hit@k, MRR@k, NDCG@k (linear gain), precision@k, context
                  precision; pinned values in tests/test_evalkit_cpu.py come from SURVEY.md 8(c)

The query embedding comes from an external model in the reference (Ollama); here the service takes an
`embed(text) -> vector` callable.  The HTTP server is the standard library's: it exists to show the
contract end to end, not to be a product server.
"""
from __future__ import annotations

import json
import math
import threading
import time
from dataclasses import dataclass, field
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Set, Tuple

DEFAULT_TOP_K = 5          # mcp_server.rs:356-358
DEFAULT_DIVERSITY = 0.3    # mcp_server.rs:359-361
MAX_TOP_K = 100            # mcp_server.rs:364


# ---------------------------------------------------------------------------------- metrics
def _top(seq: Sequence, k: int) -> Sequence:
    if k < 1:
        raise AssertionError(f"k must be >= 1, got {k}")
    return seq[:k]


def hit_rate_at_k(gold: Set[str], retrieved: Sequence[str], k: int) -> float:
    """1.0 when any gold key is among the first k retrieved keys (eval/metrics.py:18-35)"""
    return 1.0 if any(r in gold for r in _top(retrieved, k)) else 0.0


def mrr_at_k(gold: Set[str], retrieved: Sequence[str], k: int) -> float:
    """reciprocal rank of the first gold key within the first k (eval/metrics.py:38-57)"""
    for rank, r in enumerate(_top(retrieved, k), start=1):
        if r in gold:
            return 1.0 / rank
    return 0.0


def _dcg(rels: Iterable[float]) -> float:
    return sum(rel / math.log2(pos + 2) for pos, rel in enumerate(rels))


def ndcg_at_k(relevances: Sequence[int], k: int) -> float:
    """linear-gain NDCG; the ideal ordering is taken over ALL given relevances, cut at
    min(k, len) (eval/metrics.py:60-83)"""
    head = _top(relevances, k)
    if not relevances:
        return 0.0
    ideal = sorted(relevances, reverse=True)[: len(head)]
    best = _dcg(ideal)
    return _dcg(head) / best if best > 0 else 0.0


def precision_at_k(gold: Set[str], retrieved: Sequence[str], k: int) -> float:
    """distinct gold keys among the first k, over min(k, len) (eval/metrics.py:86-101)"""
    head = _top(retrieved, k)
    if not head:
        return 0.0
    return len(gold.intersection(head)) / len(head)


def context_precision(relevances: Sequence[int]) -> float:
    """share of retrieved chunks with relevance > 0 (eval/metrics.py:104-117)"""
    return sum(1 for r in relevances if r > 0) / len(relevances) if relevances else 0.0


def normalize_doc_name(name: str) -> str:
    """eval/rag_client.py:249-254"""
    return name.lower().replace(".pdf", "").strip()


def make_chunk_key(document: str, page: int) -> str:
    """eval/rag_client.py:257-262"""
    return f"{normalize_doc_name(document)}::{page}"


def gold_keys(gold_references: Sequence[dict], page_tolerance: int = 1) -> Set[str]:
    """eval/eval_runner.py:167-177: every page within the tolerance, pages >= 1 only"""
    keys = set()
    for ref in gold_references:
        for page in range(ref["page"] - page_tolerance, ref["page"] + page_tolerance + 1):
            if page >= 1:
                keys.add(make_chunk_key(ref["document"], page))
    return keys


def relevance_of(document: str, page: int, gold_references: Sequence[dict], page_tolerance: int = 1) -> int:
    """eval/eval_runner.py:179-193 with matches_gold_reference (rag_client.py:265-283)"""
    best = 0
    for ref in gold_references:
        if normalize_doc_name(document) == normalize_doc_name(ref["document"]) and abs(page - ref["page"]) <= page_tolerance:
            best = max(best, ref.get("relevance", 3))
    return best


# ---------------------------------------------------------------------------------- service
class SearchService:
    """`http_search` (mcp_server.rs:371-389) over a RagEngine: request parsing, caps, result JSON."""

    def __init__(self, engine, embed: Callable[[str], Sequence[float]], use_lexical: bool = True):
        self.engine = engine
        self.embed = embed
        self.use_lexical = use_lexical
        self._lock = threading.Lock()  # the reference holds a read lock; the Python veneer serialises

    def handle_search(self, body) -> Tuple[int, dict]:
        if not isinstance(body, dict) or not isinstance(body.get("query"), str):
            return 422, {"error": "missing field `query`"}
        top_k = body.get("top_k", DEFAULT_TOP_K)
        div = body.get("diversity_factor", DEFAULT_DIVERSITY)
        if isinstance(top_k, bool) or not isinstance(top_k, int) or top_k < 0:
            return 422, {"error": "invalid type for `top_k`: expected usize"}
        if isinstance(div, bool) or not isinstance(div, (int, float)):
            return 422, {"error": "invalid type for `diversity_factor`: expected f32"}
        top_k = min(top_k, MAX_TOP_K)
        div = min(max(float(div), 0.0), 1.0)
        try:
            with self._lock:
                q = self.embed(body["query"])
                results = self.engine.search_with_diversity(q, top_k, div, None,
                                                            query_text=body["query"] if self.use_lexical else None)
        except Exception as e:  # engine error -> INTERNAL_SERVER_ERROR (:384-387)
            return 500, {"error": f"Search error: {e}"}
        return 200, {"results": [r.to_json() for r in results]}


class _Handler(BaseHTTPRequestHandler):
    service: SearchService = None  # set by serve()

    def log_message(self, *args):  # quiet
        pass

    def _send(self, status: int, payload: dict) -> None:
        data = json.dumps(payload).encode("utf-8")
        self.send_response(status)
        self.send_header("Content-Type", "application/json")
        self.send_header("Content-Length", str(len(data)))
        self.end_headers()
        self.wfile.write(data)

    def do_POST(self):
        if self.path != "/search":
            return self._send(404, {"error": "not found"})
        if "application/json" not in (self.headers.get("Content-Type") or ""):
            return self._send(415, {"error": "expected Content-Type: application/json"})
        try:
            body = json.loads(self.rfile.read(int(self.headers.get("Content-Length") or 0)) or b"null")
        except ValueError:
            return self._send(400, {"error": "malformed JSON"})
        self._send(*self.service.handle_search(body))


def serve(service: SearchService, host: str = "127.0.0.1", port: int = 0) -> Tuple[ThreadingHTTPServer, threading.Thread]:
    """Start POST /search on a background thread; port 0 picks a free one (server.server_address)."""
    handler = type("Handler", (_Handler,), {"service": service})
    server = ThreadingHTTPServer((host, port), handler)
    thread = threading.Thread(target=server.serve_forever, daemon=True)
    thread.start()
    return server, thread


# ---------------------------------------------------------------------------------- harness
@dataclass
class RetrievedChunk:  # what eval/rag_client.py keeps of a result (:18-27, :94-104)
    chunk_id: str
    document: str
    page: int
    text: str
    score: float
    section: Optional[str] = None


def parse_search_response(data: dict) -> List[RetrievedChunk]:
    """the client's tolerant field lookup (eval/rag_client.py:94-104)"""
    out = []
    for r in data.get("results", []):
        out.append(RetrievedChunk(r.get("chunk_id", ""), r.get("document", r.get("document_name", "")),
                                  r.get("page", r.get("page_number", 0)), r.get("text", ""),
                                  r.get("score", r.get("relevance_score", 0.0)), r.get("section")))
    return out


def http_search_fn(endpoint: str, timeout: float = 60.0) -> Callable[[str, int], List[RetrievedChunk]]:
    """`search(query, top_k)` over the wire, standard library only"""
    import urllib.request

    def search(query: str, top_k: int) -> List[RetrievedChunk]:
        req = urllib.request.Request(endpoint.rstrip("/") + "/search",
                                     data=json.dumps({"query": query, "top_k": top_k}).encode("utf-8"),
                                     headers={"Content-Type": "application/json"}, method="POST")
        with urllib.request.urlopen(req, timeout=timeout) as resp:
            return parse_search_response(json.loads(resp.read()))

    return search


@dataclass
class QueryScore:
    query_id: str
    hit_rate: float
    mrr: float
    ndcg: float
    precision: float
    context_precision: float
    latency_ms: float
    retrieved_keys: List[str] = field(default_factory=list)


def evaluate(queries: Sequence[dict], search: Callable[[str, int], List[RetrievedChunk]], k: int = 5,
             page_tolerance: int = 1) -> Tuple[List[QueryScore], Dict[str, float]]:
    """Per-query scores and their means for ground-truth lines of the reference's shape
    ({"query_id", "query", "gold_references": [{"document", "page", "relevance"}]}; eval_runner.py:196-262).
    Rejection queries (is_rejection) score 1.0 when nothing is retrieved, like :229-243."""
    scores = []
    for q in queries:
        t0 = time.perf_counter()
        got = search(q["query"], k)
        ms = (time.perf_counter() - t0) * 1e3
        keys = [make_chunk_key(r.document, r.page) for r in got]
        refs = q.get("gold_references", [])
        if q.get("is_rejection"):
            scores.append(QueryScore(q.get("query_id", ""), 1.0 if not got else 0.0, 0.0, 0.0, 0.0, 0.0, ms, keys))
            continue
        gold = gold_keys(refs, page_tolerance)
        rels = [relevance_of(r.document, r.page, refs, page_tolerance) for r in got]
        scores.append(QueryScore(q.get("query_id", ""), hit_rate_at_k(gold, keys, k), mrr_at_k(gold, keys, k),
                                 ndcg_at_k(rels, k), precision_at_k(gold, keys, k), context_precision(rels), ms, keys))
    normal = [s for s, q in zip(scores, queries) if not q.get("is_rejection")]
    lat = sorted(s.latency_ms for s in scores)
    summary = {}
    if normal:
        for name in ("hit_rate", "mrr", "ndcg", "precision", "context_precision"):
            summary[f"{name}_mean"] = sum(getattr(s, name) for s in normal) / len(normal)
    if lat:
        summary["latency_p50_ms"] = lat[len(lat) // 2]
        summary["latency_p95_ms"] = lat[min(len(lat) - 1, int(0.95 * len(lat)))]
    return scores, summary
