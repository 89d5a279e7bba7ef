/*
 * rlr_engine.h -- host-side mirror of the reference's operator interface for the hot
 * path, layered on the device ABI of rlr_gpu.h.  It is what RagEngine::search /
 * search_with_diversity / get_embedding_candidates (rust-local-rag
 * src/rag_engine.rs:470-701, :717-759, :415-461) do between "query embedding obtained"
 * (:493) and "results handed back" -- same names, argument meaning and edge-case
 * behaviour -- written in C++ because this image has no Rust toolchain.  A Rust host
 * binds rlr_gpu.h directly and keeps its own copy of this logic.
 *
 * Everything that is not the hot path stays with the caller: the query string ->
 * embedding step (embeddings.rs), BM25 (`LexicalIndex::score`, passed in as
 * (row, score) pairs), the LLM reranker and chunk metadata.
 */
#ifndef RLR_ENGINE_H
#define RLR_ENGINE_H

#include "rlr_gpu.h"
#include "rlr_lexical.h"

#ifdef __cplusplus
extern "C" {
#endif

/* QueryWeights (rag_engine.rs:1846-1865): optional per-query overrides. */
typedef struct rlr_query_weights {
    int32_t has_embedding;
    float embedding;
    int32_t has_lexical;
    float lexical;
    int32_t has_reranker;
    float reranker;
    int32_t has_initial;
    float initial;
} rlr_query_weights;

/* ResolvedWeights (rag_engine.rs:1877-1896) */
typedef struct rlr_resolved_weights {
    float embedding, lexical, reranker, initial;
} rlr_resolved_weights;

/* SearchResult's numeric fields (rag_engine.rs:72-100); `row` stands in for chunk_id. */
typedef struct rlr_search_hit {
    uint64_t row;
    float score;           /* = initial_score when no reranker ran (:682) */
    float embedding_score; /* raw cosine (:689) */
    float lexical_score;   /* bm25 / max_bm25, 0 if the row is not a lexical candidate (:527-530) */
    float initial_score;   /* w_e * embedding + w_l * lexical (:531-532) */
} rlr_search_hit;

/* API-layer constants (mcp_server.rs:85-86, :359-364) */
#define RLR_MAX_TOP_K 100
#define RLR_DEFAULT_TOP_K 5
#define RLR_DEFAULT_DIVERSITY 0.3f

/* resolve_weight (rag_engine.rs:1869-1873): override if finite and in [0,1], else default */
float rlr_resolve_weight(int32_t has_override, float w, float dflt);
/* ResolvedWeights::from_query_weights (:1888-1895); defaults 0.7/0.3/0.7/0.3 or the
 * RAG_{EMBEDDING,LEXICAL,RERANKER,INITIAL_SCORE}_WEIGHT environment (:1813-1841), cached
 * on first use like the reference's OnceLock. `w` may be NULL. */
void rlr_resolve_weights(const rlr_query_weights *w, rlr_resolved_weights *out);
/* normalize (rag_engine.rs:1763-1771), host side, reference order. */
void rlr_normalize(float *v, size_t n);

/* RagEngine::search, exact-scan branch, reranker absent.
 *   query_raw/dq  raw query embedding (normalised here, :494; zip-truncated or
 *                 zero-extended to the index dim like dot_product's zip, :1778)
 *   top_k         0 is treated as 1 (:490)
 *   lex_*         LexicalIndex::score output as (row, bm25) pairs, may be empty
 *   stage         0: final results, min(top_k, N) hits (:667-698)
 *                 1: the initial_k = min(N, max(3*top_k, top_k)) candidates a reranker
 *                    would receive (:544-561)
 * Ordering: (initial_score desc, row asc), NaN last -- the build's definition of the
 * reference's unspecified tie order. */
int32_t rlr_engine_search(rlr_index *idx, const float *query_raw, uint32_t dq, uint32_t top_k,
                          const rlr_query_weights *weights, const uint64_t *lex_rows,
                          const float *lex_scores, uint32_t n_lex, int32_t stage,
                          rlr_search_hit *out, uint32_t cap, uint32_t *n_out);

/* RagEngine::search_with_diversity (:717-759): clamp lambda, lambda == 0 -> search,
 * else pool = max(3k, k+10), search(pool), MMR on the GPU. */
int32_t rlr_engine_search_with_diversity(rlr_index *idx, const float *query_raw, uint32_t dq,
                                         uint32_t top_k, float diversity_factor,
                                         const rlr_query_weights *weights, const uint64_t *lex_rows,
                                         const float *lex_scores, uint32_t n_lex,
                                         rlr_search_hit *out, uint32_t cap, uint32_t *n_out);

/* search / search_with_diversity as the reference calls them from search_documents (mcp_server.rs:81-110): with the
 * query TEXT, whose BM25 scores it blends in (`self.lexical_index.score(query, top_k.saturating_mul(5))`, :505).
 * `query_tokens`: the host's tokenize(query), space separated (as for rlr_lexical_score); `lex`: the GPU LexicalIndex
 * whose rows are the rows of `idx`.  diversity_factor == 0 (after the clamp): search(top_k) with `stage` as in
 * rlr_engine_search; otherwise search_with_diversity (stage ignored).  Same results as rlr_lexical_score followed by
 * rlr_engine_search / rlr_engine_search_with_diversity with its pairs -- but the BM25 kernels run on their own stream
 * beside the cosine scan, their result never leaves the device, and blend, ordering, cut and MMR follow in the same
 * enqueue: one host synchronisation per query (0.85 -> 0.31 ms at 100 k chunks).  Falls back to exactly those
 * two calls when the fused kernels do not cover the request (w_embedding == 0, more than 1024 candidates kept or
 * 2048 lexical pairs, a rounding-tie chain at the fetch boundary). */
int32_t rlr_engine_search_text(rlr_index *idx, rlr_lexical *lex, const float *query_raw, uint32_t dq,
                               const char *query_tokens, size_t tokens_len, uint32_t top_k, float diversity_factor,
                               int32_t stage, const rlr_query_weights *weights, rlr_search_hit *out, uint32_t cap,
                               uint32_t *n_out);

/* Additive batched entry point (the reference has no batched API; its oracle is "loop
 * search_with_diversity over the batch"): n_queries raw query embeddings,
 * no lexical candidates.  Hits of query q start at out[q * cap]; n_out[q] = their count.
 * The scan runs through the batched (matrix-core) path of rlr_search_topk and the MMR through
 * rlr_mmr_select_batch; results are identical to n_queries single calls. */
int32_t rlr_engine_search_with_diversity_batch(rlr_index *idx, const float *queries_raw, uint32_t dq,
                                               uint32_t n_queries, uint32_t top_k, float diversity_factor,
                                               const rlr_query_weights *weights, rlr_search_hit *out,
                                               uint32_t cap, uint32_t *n_out);

/* The tail of RagEngine::search once a reranker has answered (:599-700), SURVEY 8(f) row f2:
 * blended = w_reranker * relevance/max_relevance + w_initial * initial/max_initial over the
 * stage-1 candidates the reranker scored (in the reranker's order, unknown/repeated rows
 * skipped), stable sort desc, truncate to top_k, then fill from the remaining candidates by
 * initial score.  Pure host arithmetic on <= 3*top_k items.
 *   candidates / n_candidates   output of rlr_engine_search(stage = 1)
 *   rer_rows / rer_relevance    the reranker's (row, relevance) list; n_reranked == 0 = reranker
 *                               absent or failed -> pure fallback ordering
 *   out / has_reranker_out / reranker_score_out   final results (score = blended or initial),
 *                               per-result flag and raw relevance (SearchResult.reranker_score) */
int32_t rlr_engine_blend_reranked(const rlr_search_hit *candidates, uint32_t n_candidates,
                                  const uint64_t *rer_rows, const float *rer_relevance, uint32_t n_reranked,
                                  uint32_t top_k, const rlr_query_weights *weights, rlr_search_hit *out,
                                  float *reranker_score_out, int32_t *has_reranker_out, uint32_t cap,
                                  uint32_t *n_out);

/* RagEngine::get_embedding_candidates (:415-461), `None` arm: top `count` by cosine. */
int32_t rlr_engine_embedding_candidates(rlr_index *idx, const float *query_raw, uint32_t dq,
                                        uint32_t count, uint64_t *rows_out, float *scores_out,
                                        uint32_t *n_out);

/* ---- the same entry points over a corpus sharded across several GPUs in ONE process (rlr_multi, include/rlr_gpu.h) ----
 * The reference is a single-process server whose surface is RagEngine::search / search_with_diversity
 * (rag_engine.rs:470-475, :717-723, called from mcp_server.rs:89-93); a corpus that does not fit one GPU (BASELINE
 * config 4: 100 M x 768 f32) is held by an rlr_multi and searched through these.  Same arguments, defaults, fall-backs
 * and results as the rlr_engine_* functions above over one index holding every row -- the host logic is the same code
 * (csrc/engine_host.h) on the sharded primitives: per-shard scan + exchange + merge for the candidates
 * (rlr_multi_search_topk: host merge or RCCL all-gather, rlr_multi_set_exchange), reference-order cosines of the
 * lexical rows from the shards that own them, and MMR with the on-fabric winner-row exchange
 * (rlr_multi_mmr_select(_batch)).  Row numbers are global.  `lex` of _search_text is ONE GPU LexicalIndex over the
 * global rows (BM25 postings are small next to the embeddings; they stay on one device). */
int32_t rlr_multi_engine_search(rlr_multi *m, const float *query_raw, uint32_t dq, uint32_t top_k,
                                const rlr_query_weights *weights, const uint64_t *lex_rows, const float *lex_scores,
                                uint32_t n_lex, int32_t stage, rlr_search_hit *out, uint32_t cap, uint32_t *n_out);
int32_t rlr_multi_engine_search_with_diversity(rlr_multi *m, const float *query_raw, uint32_t dq, uint32_t top_k,
                                               float diversity_factor, const rlr_query_weights *weights,
                                               const uint64_t *lex_rows, const float *lex_scores, uint32_t n_lex,
                                               rlr_search_hit *out, uint32_t cap, uint32_t *n_out);
int32_t rlr_multi_engine_search_text(rlr_multi *m, rlr_lexical *lex, const float *query_raw, uint32_t dq,
                                     const char *query_tokens, size_t tokens_len, uint32_t top_k, float diversity_factor,
                                     int32_t stage, const rlr_query_weights *weights, rlr_search_hit *out, uint32_t cap,
                                     uint32_t *n_out);
/* BASELINE config 5's shape (1024 queries, top-100, lambda 0.7 over a sharded corpus): one batched top-k per shard +
 * exchange, per-query pools, one batched cross-shard MMR. */
int32_t rlr_multi_engine_search_with_diversity_batch(rlr_multi *m, const float *queries_raw, uint32_t dq,
                                                     uint32_t n_queries, uint32_t top_k, float diversity_factor,
                                                     const rlr_query_weights *weights, rlr_search_hit *out,
                                                     uint32_t cap, uint32_t *n_out);
int32_t rlr_multi_engine_embedding_candidates(rlr_multi *m, const float *query_raw, uint32_t dq, uint32_t count,
                                              uint64_t *rows_out, float *scores_out, uint32_t *n_out);

/* ---- corpus file at scale ---------------------------------------------------------
 * `chunks_{model}.json` (PersistedState, rag_engine.rs:1478-1499; read whole with serde_json :1555-1557) read
 * in one streaming pass over the memory-mapped file: the embedding arrays go straight into a dense row-major f32
 * matrix (row r = the r-th chunk of the file; decimal -> binary64 correctly rounded -> binary32, serde_json's f32
 * path; null = a non-finite value; shorter / longer arrays are zero-extended / truncated to `dim` like
 * dot_product's zip :1778), everything else is copied verbatim into `meta_json`, the same document with every
 * embedding array replaced by [] -- id, text, metadata, document_hashes, version, model: small enough for the
 * host's own JSON library.  Buffers are malloc'ed; release them with rlr_json_free_corpus. */
typedef struct rlr_json_corpus {
    float *rows;       /* n_rows x dim */
    uint64_t n_rows;
    uint32_t dim;
    char *meta_json;   /* NUL-terminated */
    uint64_t meta_len;
} rlr_json_corpus;
int32_t rlr_json_load_corpus(const char *path, uint32_t dim, rlr_json_corpus *out);
void rlr_json_free_corpus(rlr_json_corpus *c);
/* The same, uploaded: replaces all rows of `idx` (rlr_index_upload; normalize_on_device = 1 reproduces the
 * reference's re-normalise-on-load :1678-1680 on the GPU).  `meta_out` (nullable) receives the metadata document
 * (rows == NULL there: they live in HBM). */
int32_t rlr_index_load_json(rlr_index *idx, const char *path, int32_t normalize_on_device, rlr_json_corpus *meta_out);
/* One embedding as save_to_disk writes it (:1477-1518, serde_json pretty printer): "[", one value per line at
 * indent + 2 spaces -- the shortest decimal that reads back as the same binary32, non-finite as null -- and "]" at
 * `indent`.  Returns the bytes needed; the text was written (no terminating NUL) only if that is <= cap. */
uint64_t rlr_json_format_embedding(const float *v, uint32_t dim, uint32_t indent, char *out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* RLR_ENGINE_H */
