"""Quick tour of the Python veneer over librlr_gpu.so (needs an MI355X; the library has no CPU path).

    python examples/quickstart.py

What the reference does in `RagEngine::add_document` / `search_with_diversity` / `search_documents`
(src/rag_engine.rs, src/mcp_server.rs) with the embeddings in a HashMap and scalar loops, with the embeddings in HBM.
The query embedding comes from an external model in the reference; random vectors stand in for it here."""
import importlib
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rlr = importlib.import_module("rust-local-rag_amd")

dim = 768
rng = np.random.default_rng(0)
eng = rlr.RagEngine(dim)                                   # accelerate="image" / "q8": optional nomination copies

# -- ingest two documents (chunk texts + their embeddings; rows are normalised on the GPU like :358-359) --------
for name, n in (("fox.pdf", 400), ("dog.pdf", 600)):
    texts = [f"{name} chunk {i}: the quick brown fox jumps over the lazy dog number {i % 7}" for i in range(n)]
    eng.add_document(name, texts, rng.standard_normal((n, dim)).astype(np.float32), pages=[1 + i // 4 for i in range(n)])
print("chunks resident in HBM:", len(eng))

# -- hybrid search: cosine scan on the GPU + BM25 of the query text on the GPU, MMR diversification ------------
q = rng.standard_normal(dim).astype(np.float32)
hits = eng.search_documents(rlr.SearchRequest(query_embedding=q, query="quick fox number 3", top_k=5))
print(rlr.format_search_results(hits)[:400], "...")

# -- per-query weights (rag_engine.rs:1846-1896), plain search, candidates for a reranker --------------------------
hits = eng.search(q, 3, weights=rlr.QueryWeights(embedding=0.9, lexical=0.1), query_text="lazy dog")
print([(h.document, h.page_number, round(h.score, 4)) for h in hits])
stage1 = eng.search(q, 3, stage=1)                         # the 3 * top_k candidates a reranker would see
final = eng.finish_with_reranker(stage1, [(stage1[2].chunk_id, 0.9), (stage1[0].chunk_id, 0.4)], 3)
print([h.chunk_id[:8] for h in final], [h.reranker_score for h in final])

# -- replace a document, persist, reload (chunks_{model}.json, reference format; binary side-car cache) -----------
eng.add_document("fox.pdf", ["a single replacement chunk"], rng.standard_normal((1, dim)).astype(np.float32))
with tempfile.TemporaryDirectory() as d:
    path = rlr.save_to_disk(eng, d, "nomic-embed-text", document_hashes={"fox.pdf": "h1", "dog.pdf": "h2"})
    eng2 = rlr.RagEngine(dim, accelerate="q8")            # same results, the scan streams a quarter of the bytes
    rep = rlr.load_from_disk(eng2, d, "nomic-embed-text", use_sidecar=True)
    print("reloaded", rep.n_chunks, "chunks from", os.path.basename(rep.source))
    a = [h.chunk_id for h in eng.search(q, 5)]
    b = [h.chunk_id for h in eng2.search(q, 5)]
    print("same top-5 after the round trip:", a == b)
    eng2.close()
eng.close()
