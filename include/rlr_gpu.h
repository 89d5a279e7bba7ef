/*
 * rlr_gpu.h -- C ABI of the MI355X-native search_documents hot path.
 *
 * This is the drop-in boundary: the entry points below are exactly what a Rust
 * `extern "C"` block in rust-local-rag's src/rag_engine.rs would bind (see
 * INTEGRATION.md for the binding and the patch).  The reference has no FFI /
 * plugin seam of its own for this path (the scan is inline in
 * `RagEngine::search`), so every function cites the reference lines it
 * replaces.  Paths are relative to the reference repository root.
 *
 * Conventions
 *   - plain C: opaque handle, pointers and sizes only; no exceptions cross it.
 *   - every function returns an int32 status: RLR_OK (0) or a negative code;
 *     rlr_last_error() returns a thread-local message for the last failure.
 *   - the caller owns every buffer it passes; the library owns only what sits
 *     behind the opaque handle.
 *   - rows are dense, row-major, `dim` elements each; row numbers are the
 *     position in upload/append order ("row" <-> chunk_id table stays with the
 *     host, src/rag_engine.rs:105).
 *   - thread-safety mirrors the reference's tokio RwLock (mcp_server.rs:89,
 *     worker.rs:397-399): search / score / fetch / mmr calls may run
 *     concurrently from any number of OS threads; mutators (upload, append,
 *     delete, fill, reserve, enable_batch_image, destroy) need external exclusion.
 *   - there is NO CPU fallback inside the library: with no usable GPU every
 *     compute entry point returns RLR_E_NO_DEVICE.
 *   - every entry point returns when its results are in the caller's buffers.
 *     How it waits for the device is the library's business (RLR_WAIT, see
 *     INTEGRATION.md section 6): by default it polls a completion word the last
 *     kernel stores behind its results in pinned host memory, and verifies a
 *     checksum over them, instead of waiting for the stream's completion signal.
 */
#ifndef RLR_GPU_H
#define RLR_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLR_VERSION 121 /* 0.1.21: + bandwidth probe, n_batches_without_image, n_mmr_host_bounces (both structs grew at the end) */

typedef struct rlr_index rlr_index;

enum rlr_status {
    RLR_OK = 0,
    RLR_E_INVALID = -1,   /* bad argument */
    RLR_E_NO_DEVICE = -2, /* no usable HIP device / device id out of range */
    RLR_E_HIP = -3,       /* a HIP runtime call failed (message in rlr_last_error) */
    RLR_E_OOM = -4,       /* device or pinned-host allocation failed */
    RLR_E_RANGE = -5,     /* a row number is >= the index size */
    RLR_E_INTERNAL = -6
};

enum rlr_dtype {
    RLR_F32 = 0, /* rows stored as IEEE binary32 (the reference's Vec<f32>, rag_engine.rs:51) */
    RLR_F16 = 1  /* rows rounded to binary16 after normalisation; arithmetic on the widened values */
};

/* ---- library ---------------------------------------------------------- */
int32_t rlr_version(void);
/* number of visible HIP devices (0 when the runtime reports none) */
int32_t rlr_device_count(void);
const char *rlr_last_error(void);
/* guard band the approximate scan uses for a given dim: an upper
 * bound on |wavefront-order dot - reference-order dot| for unit-norm operands. */
float rlr_default_guard_eps(uint32_t dim);

/* ---- index lifetime and mutation --------------------------------------- */
/* replaces: the `chunks: HashMap<String, DocumentChunk>` embedding storage
 * (rag_engine.rs:46-59, :105) by one dense row-major matrix in HBM. */
int32_t rlr_index_create(uint32_t dim, int32_t dtype, int32_t device_id, rlr_index **out);
int32_t rlr_index_destroy(rlr_index *idx);
int32_t rlr_index_info(const rlr_index *idx, uint64_t *n_rows, uint32_t *dim, int32_t *dtype,
                       int32_t *device_id);
int32_t rlr_index_reserve(rlr_index *idx, uint64_t n_rows);

/* Replace all rows.  rows: n_rows x dim f32, host memory.
 * normalize_on_device != 0 applies the reference `normalize` (rag_engine.rs:1763-1771:
 * sequential sum of squares, skip if <= 1e-20, true division by sqrtf) to every row on
 * the GPU, bit-identically -- the bulk re-normalise-on-load site rag_engine.rs:1675-1680.
 * With 0 the rows are stored as given (already normalised by the host, :358-359). */
int32_t rlr_index_upload(rlr_index *idx, const float *rows, uint64_t n_rows,
                         int32_t normalize_on_device);
/* Append rows (site rag_engine.rs:379-384); *first_row_out = row number of rows[0]. */
int32_t rlr_index_append(rlr_index *idx, const float *rows, uint64_t n_rows,
                         int32_t normalize_on_device, uint64_t *first_row_out);
/* Delete rows (site `chunks.retain(..)` rag_engine.rs:347-348, :265-266).  Stable
 * compaction: surviving rows keep their relative order and are renumbered densely, i.e.
 * new_row = old_row - |{deleted d : d < old_row}|.  Duplicates in `rows` are ignored. */
int32_t rlr_index_delete_rows(rlr_index *idx, const uint64_t *rows, uint64_t n);
/* Fill the index with the deterministic synthetic corpus of SURVEY.md 8(d) without
 * crossing PCIe: rows [row0, row0+n_rows) of stream `seed` (integer-only generator,
 * bit-identical to oracle/rlr_oracle.c:rlr_o_synth_rows), reference-normalised, then
 * rounded to the index dtype.  Benchmark / test support.
 * n_clusters: low 30 bits = number of cluster centres (0: iid rows); bit 31: "tight" clusters (the rows of a
 * cluster are near-copies of each other, cosine ~0.999: a dense top of the ranking); bit 30: the last
 * n_rows / 100 rows repeat the first n_rows / 100 (exact duplicate chunks, as re-ingesting a document
 * without deleting it first leaves them, rag_engine.rs:347-384). */
int32_t rlr_index_fill_synthetic(rlr_index *idx, uint64_t n_rows, uint64_t row0, uint64_t seed,
                                 uint32_t n_clusters);

/* Keep (enable != 0) or drop a binary16 "nomination image" of the rows laid out for the batched
 * matrix-core path: [tile of 256 rows][K-chunk][wave][MFMA fragment], every fragment load lane-
 * linear and a tile contiguous in HBM.  Costs dim * 2 bytes per row; kept in sync by upload /
 * append / delete / fill.  Results are unchanged (the image only nominates; nominated rows are
 * re-scored from the row-major master copy); batched throughput roughly doubles (rlr_profile's
 * n_batches_without_image counts the batches that would have gained).  It is never switched on
 * behind the caller's back: + 50 % of an f32 index's HBM is the caller's decision.  dim % 64 == 0.
 * enable = 3 (bit 1) additionally lets SINGLE queries over f32 rows nominate from the image: the
 * HBM-bound scan then streams 2 bytes per element instead of 4 (10 M x 768: 2.5 ms instead of 4.6 ms
 * per query), the guard band widens to the binary16 rounding bound, the re-score is unchanged --
 * identical results.
 * enable bit 2 (value 4, alone or with the others) keeps an 8-bit copy with per-row scales for single
 * queries instead: dim + 4 bytes per row, a quarter of the scan bytes (10 M x 768: 1.4 ms per query); the
 * band comes from the exact per-row quantisation error norms kept at build time (Cauchy-Schwarz), so the
 * results are still identical.  dim % 16 == 0, dim <= 2048 (f32 or binary16 rows). */
int32_t rlr_index_enable_batch_image(rlr_index *idx, int32_t enable);

/* ---- the hot path ------------------------------------------------------- */
/* Brute-force cosine scan + top-k.
 * replaces: the exact-scan branch of RagEngine::search, rag_engine.rs:496-503 (ids = all
 * chunks) + :524-545 (dot per chunk, sort desc, take) and get_embedding_candidates
 * :431-445.
 *   queries   n_queries x dim f32, host memory, ALREADY normalised by the caller in
 *             reference order (rag_engine.rs:494) -- 3 KB, stays on the host so the staged
 *             query is bit-identical.
 *   k         results wanted per query (clamped to the index size).
 *   guard_eps half-width of the guard band FOR UNIT-NORM OPERANDS; < 0 selects rlr_default_guard_eps(dim).
 *             The library widens it by |row|_max * |query|_max whenever that product exceeds 1 (rows stored
 *             with normalize_on_device = 0 have their largest norm recorded at upload / append; the queries'
 *             norms are taken per call), so unnormalised data keeps the exactness guarantee.
 *   rows_out  n_queries x k row numbers, cos_out n_queries x k scores, n_out[q] = count.
 * Result per query: the k rows with the largest reference-order dot product
 * (dot_product, rag_engine.rs:1777-1779), scores bit-identical to it, ordered by
 * (score desc, row asc); NaN scores order last.  The wavefront-order scan only
 * nominates candidates (everything within the guard band of the k-th score); the
 * nominated rows are re-scored on the GPU in strict reference order before the final
 * ordering, so the band never leaks into the result.
 * Batches on dims that are a multiple of 128 take a batched pipeline when a cost model says it is cheaper
 * than n_queries single scans (from 2 queries at 10 M rows, from ~13 at a few thousand rows; RLR_BATCH_MIN=n
 * forces a threshold).  2..8 queries over f32 rows share ONE wavefront-order pass over the rows
 * (scan_multi_kernel); larger batches take the matrix-core path: Q x Corpus^T by v_mfma_f32_16x16x32_f16 on binary16-rounded operands nominates
 * candidates (rigorous band), the same reference-order re-score finishes; the
 * results are identical to looping single queries. */
int32_t rlr_search_topk(rlr_index *idx, const float *queries, uint32_t n_queries, uint32_t k,
                        float guard_eps, uint64_t *rows_out, float *cos_out, uint32_t *n_out);

/* Reference-order dot product of one query with the listed rows (the <= 5*top_k lexical
 * candidates of rag_engine.rs:505-509, whose embedding score the hybrid blend needs).
 * replaces: dot_product(&query_embedding, &chunk.embedding) rag_engine.rs:526. */
int32_t rlr_score_rows(rlr_index *idx, const float *query, const uint64_t *rows, uint32_t n,
                       float *cos_out);

/* Copy rows back to the host as f32 (exact values; fp16 rows are widened).
 * replaces: the `chunk.embedding.clone()` re-lookups rag_engine.rs:559, :742-753. */
int32_t rlr_fetch_rows(rlr_index *idx, const uint64_t *rows, uint32_t n, float *out);

/* Maximal-marginal-relevance selection over a candidate pool that lives in the index.
 * replaces: RagEngine::mmr_diversify, rag_engine.rs:767-839 (greedy loop :788-835).
 *   pool_rows / pool_scores  P candidates in the order `search` returned them (the
 *             reference's visiting order; swap_remove perturbs it exactly as :783, :825)
 *   k         results wanted (k == 0 still yields the first candidate, as :782-785)
 *   lambda    diversity factor, already clamped by the caller (:725)
 *   order_out indices into the pool in pick order; mmr_out (nullable) the MMR value of
 *             each pick (NaN for the first); *n_out = number of picks.
 * Pairwise similarities are reference-order dot products computed on the GPU, so the
 * pick sequence and MMR values are bit-identical to the reference loop. */
int32_t rlr_mmr_select(rlr_index *idx, const uint64_t *pool_rows, const float *pool_scores,
                       uint32_t P, uint32_t k, float lambda, uint32_t *order_out, float *mmr_out,
                       uint32_t *n_out);

/* The same selection for a batch of queries in one set of launches: pools are P-strided
 * (pool_rows / pool_scores / order_out / mmr_out are n_queries x P), pool_sizes[q] <= P <= 1024
 * candidates are valid for query q.  One wavefront per query runs the greedy loop, all queries
 * concurrently.  Per-query results are identical to rlr_mmr_select. */
int32_t rlr_mmr_select_batch(rlr_index *idx, const uint64_t *pool_rows, const float *pool_scores,
                             const uint32_t *pool_sizes, uint32_t n_queries, uint32_t P, uint32_t k,
                             float lambda, uint32_t *order_out, float *mmr_out, uint32_t *n_out);

/* search -> pool -> MMR in ONE enqueue on one stream with ONE host synchronisation: the path of
 * RagEngine::search_with_diversity (rag_engine.rs:717-759) for a query without lexical candidates --
 * `search(pool)` :735 with combined = w_e*cos + w_l*0 (:531-532), candidate order (combined desc, row asc)
 * :543, cut to `pool` :544/:734, then `mmr_diversify` :756 -- with the pool never leaving the device
 * (scan -> select -> re-score -> pool order -> gather -> Gram -> greedy -> results in pinned memory).
 *   query        normalised, dim floats (as for rlr_search_topk)
 *   pool         candidates `search` keeps (the reference's max(3*top_k, top_k+10)), <= 1024
 *   k, lambda    as rlr_mmr_select
 *   w_embedding, w_lexical   the resolved weights (w_embedding > 0)
 *   rows_out / cos_out / score_out   min(max(k,1), pool) entries each, in pick order; score = the combined
 *                (relevance) score the reference reports, cos = embedding_score
 *   *fallback    != 0: nothing was written; take rlr_search_topk + rlr_mmr_select instead.  Set when the
 *                guard band overflowed (massive exact ties), when distinct cosines round to one combined score
 *                in a chain that reaches the last fetched row (the boundary rule of rag_engine.rs:543 needs a
 *                wider fetch), or for arguments outside the fused kernels (pool > 1024, w_embedding <= 0).
 * Results are identical to the two-call path (tests/test_gpu_parity.py::test_search_diverse_*). */
int32_t rlr_search_diverse(rlr_index *idx, const float *query, uint32_t pool, uint32_t k, float lambda,
                           float w_embedding, float w_lexical, float guard_eps, uint64_t *rows_out,
                           float *cos_out, float *score_out, uint32_t *n_out, int32_t *fallback);

/* The same single enqueue for the HYBRID search the reference runs whenever the query has text (rag_engine.rs:505-561:
 * `lexical_index.score(query, top_k * 5)`, then combined = w_e*cos + w_l*(lex / max_lex) over every chunk, order
 * (combined desc, row asc), cut to `need`), optionally followed by mmr_diversify (:756) -- scan -> select -> re-score ->
 * cosines of the lexical rows -> blend + order + cut -> [Gram -> greedy] -> results in pinned memory, one
 * synchronisation.  Candidates are the need + n_lex + 8 best rows by cosine united with the lexical rows: a row
 * without a lexical score is ordered by its cosine alone, so nothing else can reach the first `need`.
 *   need         candidates the search keeps: min(top_k, 3*top_k) for a plain search (:544, :667-698), the pool
 *                max(3*top_k, top_k+10) under diversification (:734); <= 1024
 *   diversify    0: return the first `need` candidates in order (search); != 0: MMR-select k of them with `lambda`
 *   lex_rows / lex_scores   the BM25 pairs as rows: ascending, unique, inside the index; n_lex <= 2048 and
 *                need + 2*n_lex + 8 <= 4096
 *   max_lex      max(lexical scores, f32::EPSILON) as the reference computes it (:515-519)
 *   rows_out / cos_out / score_out / lex_out   `need` entries (diversify: min(max(k,1), need)): row, embedding_score,
 *                combined score, normalised lexical score (0 for rows without one)
 *   *fallback    != 0: nothing was written, take the host path (same conditions as rlr_search_diverse, or sizes
 *                outside the limits above).
 * Results are identical to the host path of csrc/engine.cpp (tests: test_engine_hybrid_*, fuzz_engine). */
int32_t rlr_search_hybrid(rlr_index *idx, const float *query, uint32_t need, uint32_t k, float lambda, int32_t diversify,
                          float w_embedding, float w_lexical, const uint64_t *lex_rows, const float *lex_scores,
                          uint32_t n_lex, float max_lex, float guard_eps, uint64_t *rows_out, float *cos_out,
                          float *score_out, float *lex_out, uint32_t *n_out, int32_t *fallback);

/* ---- device-resident variant (multi-GPU sharding) ------- */
/* Same search, but the per-query result stays in device memory so the caller can hand it
 * to an RCCL all-gather without a host round trip.
 *   d_packed_out  device pointer, n_queries x k uint64: (ordered-score-bits << 32) |
 *                 (0xFFFFFFFF - local_row); descending u64 order == (score desc, row asc).
 *                 Unused tail entries are 0.
 *   stream        hipStream_t the caller will consume the result on (may be NULL = the
 *                 null stream); the library makes its work visible to that stream. */
int32_t rlr_search_topk_device(rlr_index *idx, const float *queries, uint32_t n_queries,
                               uint32_t k, float guard_eps, void *d_packed_out, void *stream);
/* The winner-row exchange of sharded MMR: rows of this shard as f32 values in
 * DEVICE memory (`d_out`, n x dim floats, dense), ready to be handed to an RCCL all-to-all.
 * Returns after the gather has finished.  replaces: the embedding re-lookup rag_engine.rs:742-753
 * for rows that live on this GPU. */
int32_t rlr_fetch_rows_device(rlr_index *idx, const uint64_t *rows, uint32_t n, void *d_out);
/* The same gather without the widening: rows as the index stores them (rlr_index_row_bytes bytes each: dim elements
 * of the index' dtype, padded to 16 bytes), dense, in DEVICE memory -- the send side of the one-process winner-row
 * exchange (binary16 rows travel as 2 bytes per element).  Returns after the gather has finished. */
int32_t rlr_gather_rows_device(rlr_index *idx, const uint64_t *rows, uint32_t n, void *d_out);
int32_t rlr_index_row_bytes(const rlr_index *idx, uint32_t *bytes_out);
/* rlr_mmr_select_batch for pools whose row VALUES are already in device memory (after the
 * exchange): d_values is n_queries x P x dim f32 on idx's device, complete before the call
 * (synchronise the stream that produced it).  `idx` supplies the device, dim and workspace; its
 * rows are not read.  Everything else as rlr_mmr_select_batch. */
/* rlr_mmr_select_batch for pools whose rows sit in a STAGED matrix in device memory instead of the index: d_staged =
 * n_staged raw rows in idx's dtype and row pitch (what rlr_gather_rows_device writes; e.g. the receive buffer of the
 * winner-row exchange), pool_slots[q * P + j] = the staged row that is candidate j of pool q.  `idx` supplies device,
 * dim, dtype and workspace; its own rows are not read.  n_queries == 1 takes pools up to 4096 candidates. */
int32_t rlr_mmr_select_staged(rlr_index *idx, const void *d_staged, uint64_t n_staged, const uint64_t *pool_slots,
                              const float *pool_scores, const uint32_t *pool_sizes, uint32_t n_queries, uint32_t P,
                              uint32_t k, float lambda, uint32_t *order_out, float *mmr_out, uint32_t *n_out);
int32_t rlr_mmr_select_values(rlr_index *idx, const void *d_values, const float *pool_scores,
                              const uint32_t *pool_sizes, uint32_t n_queries, uint32_t P, uint32_t k,
                              float lambda, uint32_t *order_out, float *mmr_out, uint32_t *n_out);
/* The same search split in two so the caller's collective can be queued behind the scan without a host
 * round trip (the sharded step: begin -> all-gather -> rlr_merge_topk -> end).  begin() enqueues the
 * pipelines on the caller's `stream` itself, so whatever is queued there next is ordered behind them;
 * `*ticket_out` keeps the search context until end().  end() joins and reports in *n_overflow_out how many queries overflowed their guard
 * band (massive exact ties): their slots of d_packed_out start with the all-ones word instead of results --
 * rlr_merge_topk reports such a query with n_out[q] = 0xFFFFFFFF on every rank -- and the caller re-runs the
 * step with rlr_search_topk_device.  Batched / profiled / k > rows calls run synchronously inside begin() and hand
 * back a NULL ticket (end() is then a no-op). */
int32_t rlr_search_topk_device_begin(rlr_index *idx, const float *queries, uint32_t n_queries, uint32_t k,
                                     float guard_eps, void *d_packed_out, void *stream, void **ticket_out);
int32_t rlr_search_topk_device_end(rlr_index *idx, void *ticket, uint32_t *n_overflow_out);
/* The exchange step's merge: `d_gathered` is the all-gathered buffer, world x n_queries x k packed
 * u64 (rank-major), on device `device_id`; bases[r] = first global row of rank r's shard
 * (ascending).  Emits per query the global top-k as (global row, score) to host buffers, ordered
 * (score desc, global row asc); n_out[q] = number of valid results (<= k; unused slots get row
 * ~0 and NaN), or 0xFFFFFFFF when a shard marked the query invalid (rlr_search_topk_device_begin).  world <= 16, world * k <= 8192.  Runs on `stream` (hipStream_t or NULL) and
 * returns after it has finished. */
int32_t rlr_merge_topk(int32_t device_id, const void *d_gathered, uint32_t world, uint32_t n_queries,
                       uint32_t k, const uint64_t *bases, uint64_t *rows_out, float *cos_out,
                       uint32_t *n_out, void *stream);
/* helpers for the packed format (host side) */
uint64_t rlr_pack_result(float score, uint32_t row);
void rlr_unpack_result(uint64_t packed, float *score, uint32_t *row);

/* ---- one process, several GPUs ------------------------------------------------------------ */
/* The reference is a single-process server; this handle lets such a host drive all GPUs of a node
 * without torch / one-process-per-GPU: rows are split into contiguous ranges (shard g holds rows
 * [g*ceil(N/G), ...)), every search runs on all shards concurrently (one persistent host thread per
 * shard, parked between calls) and the per-shard top-k lists (k x 8 bytes each) are merged on the host or, with
 * rlr_multi_set_exchange(m, 1), all-gathered over RCCL and merged on the device.  Results are identical
 * to a single index over the same rows.  device_ids may repeat (several shards on one GPU).
 * The benchmark contract's multi-GPU path is the one-process-per-GPU / RCCL variant
 * (rlr_search_topk_device + rlr_merge_topk, rust-local-rag_amd/sharded.py). */
typedef struct rlr_multi rlr_multi;
int32_t rlr_multi_create(uint32_t dim, int32_t dtype, int32_t n_devices, const int32_t *device_ids,
                         rlr_multi **out);
int32_t rlr_multi_destroy(rlr_multi *m);
int32_t rlr_multi_info(const rlr_multi *m, uint64_t *n_rows, uint32_t *n_shards);
/* How rlr_multi_search_topk exchanges the per-shard partial top-k lists:
 *   0  host merge (default): each shard's k x (row, score) list returns to the host, k-way merge there;
 *   1  RCCL: the lists stay in device memory (rlr_search_topk_device), one ncclAllGather of
 *      n_queries x k x 8 bytes per shard over xGMI (group call, single process: ncclCommInitAll), then
 *      merge_topk_kernel on the first device (rlr_merge_topk).  Needs one shard per device (RLR_E_INVALID
 *      otherwise) and a loadable librccl.so (dlopen on first use; RLR_E_NO_DEVICE otherwise).  A call whose
 *      shape is outside the merge kernel (more than 16 shards, shards x k > 8192) or whose guard band
 *      overflowed on some shard is answered through the host merge instead.  Results are identical in both
 *      modes.  Mutator-class call: not concurrent with searches. */
int32_t rlr_multi_set_exchange(rlr_multi *m, int32_t mode);
/* replace all rows (host memory, n_rows x dim f32), sharded by contiguous ranges */
int32_t rlr_multi_upload(rlr_multi *m, const float *rows, uint64_t n_rows, int32_t normalize_on_device);
int32_t rlr_multi_fill_synthetic(rlr_multi *m, uint64_t n_rows, uint64_t seed, uint32_t n_clusters);
/* rlr_index_enable_batch_image on every shard (the optional nomination copies: same flags, same guarantees) */
int32_t rlr_multi_enable_batch_image(rlr_multi *m, int32_t enable);
/* same contract as rlr_search_topk, row numbers are global */
int32_t rlr_multi_search_topk(rlr_multi *m, const float *queries, uint32_t n_queries, uint32_t k,
                              float guard_eps, uint64_t *rows_out, float *cos_out, uint32_t *n_out);
int32_t rlr_multi_score_rows(rlr_multi *m, const float *query, const uint64_t *rows, uint32_t n, float *cos_out);
int32_t rlr_multi_fetch_rows(rlr_multi *m, const uint64_t *rows, uint32_t n, float *out);
/* MMR over pools whose rows live on different shards (SURVEY.md 8(e) "MMR on sharded data"; replaces the embedding
 * re-lookup + mmr_diversify of rag_engine.rs:742-756 for a sharded corpus).  The winner-row exchange stays on the
 * fabric: every shard gathers the pool rows it owns on its own device (raw rows: binary16 stays binary16), one
 * device-to-device copy per (source shard, owner) pair (hipMemcpyPeerAsync over xGMI; no host bounce) puts them into
 * the receive buffer of the GPU that owns the query -- query q is diversified on shard q mod n_shards -- and that GPU
 * runs the Gram + greedy kernels over its buffer.  Same results as rlr_mmr_select(_batch) over one index holding every
 * row.  Arguments as rlr_mmr_select / rlr_mmr_select_batch with GLOBAL row numbers; one pool: P <= 4096, batch:
 * P <= 1024.  Concurrent callers are served (up to four exchanges in flight, the rest wait). */
int32_t rlr_multi_mmr_select(rlr_multi *m, const uint64_t *pool_rows, const float *pool_scores, uint32_t P,
                             uint32_t k, float lambda, uint32_t *order_out, float *mmr_out, uint32_t *n_out);
int32_t rlr_multi_mmr_select_batch(rlr_multi *m, const uint64_t *pool_rows, const float *pool_scores,
                                   const uint32_t *pool_sizes, uint32_t n_queries, uint32_t P, uint32_t k, float lambda,
                                   uint32_t *order_out, float *mmr_out, uint32_t *n_out);
/* Which path the calls on this handle took (no reference counterpart; a multi-GPU run explains its own numbers):
 * rlr_multi_search_topk calls served by the RCCL exchange / by the host merge / RCCL calls that fell through to the
 * host merge (more than 16 shards, shards x k > 8192, a shard's guard band overflowed), wall time of the RCCL form
 * (local pipelines + all-gather + merge kernel), and the cross-shard MMR exchanges with their bytes and wall time
 * (gather + device-to-device copies, without the Gram / greedy kernels). */
typedef struct rlr_multi_stats_t {
    uint64_t n_topk_rccl, n_topk_host_merge, n_topk_rccl_fell_back;
    double topk_rccl_ms;
    uint64_t n_mmr_exchanges, mmr_exchange_bytes;
    double mmr_exchange_ms;
    uint64_t n_mmr_host_bounces; /* winner-row pieces that went through host memory because the peer copy failed */
} rlr_multi_stats_t;
int32_t rlr_multi_stats(rlr_multi *m, rlr_multi_stats_t *out, int32_t reset);

/* ---- measurement hooks --------------------------------------------------- */
typedef struct rlr_profile {
    uint64_t n_searches;   /* rlr_search_topk* calls (queries, not batches) since reset */
    uint64_t n_scan_launches;
    double scan_ms;        /* sum of HIP-event durations of the scan kernel, on its stream */
    double select_ms;      /* tail stage 1: bin search + collect + reference-order re-score of what it finds (or the
                            * digit-2 histogram of a crowded bin); in the split form: histogram / threshold / collect */
    double rescore_ms;     /* tail stage 2: final sort + emit (after a digit-2 histogram: collect + re-score + sort); in
                            * the split form: reference-order re-score + final sort */
    double total_ms;       /* first launch -> results ready, per call, summed */
    uint64_t scan_bytes;   /* algorithmic bytes the scan launches covered (rows*dim*elem) */
    uint64_t n_candidates; /* rows nominated by the guard band, summed */
    uint64_t n_retries;    /* band overflow retries */
    /* batched (MFMA) path */
    uint64_t n_batches;        /* batched pipelines run */
    uint64_t n_batch_queries;  /* queries they carried */
    double batch_gemm_ms;      /* sum of HIP-event durations of the GEMM nomination launches */
    double batch_other_ms;     /* query prep + sample select + per-query finish */
    uint64_t batch_gemm_bytes; /* algorithmic bytes: the operand the GEMM streams (rows * dim * 2 B over the
                                * nomination image or binary16 rows, * 4 B over f32 rows), ONCE per batch: the
                                * query blocks of one row tile share it through the XCD's L2 */
    double batch_gemm_flops;   /* 2 * queries * rows * dim */
    uint64_t n_batch_fallbacks;/* queries re-run through the single-query pipeline */
    /* the dominant launch of a batch alone: the filtered main pass over rows [sample end, n) */
    double batch_main_ms;      /* sum of its HIP-event durations */
    uint64_t batch_main_bytes; /* rows it covered * dim * operand bytes */
    double batch_main_flops;   /* 2 * queries * rows it covered * dim */
    /* MMR (rlr_mmr_select*, rlr_engine_search_with_diversity*) */
    uint64_t n_mmr;            /* queries diversified */
    double mmr_ms;             /* gather + Gram + greedy kernels, HIP events on their stream, summed per call */
    /* batches of >= 16 queries that streamed the f32 ROWS through the matrix cores because the index keeps no
     * nomination image (rlr_index_enable_batch_image) although its shape allows one: the same results at about half
     * the batched throughput (10 M x 768, 256 queries: 31 k instead of 59 k queries/s).  A host that sees this count
     * grow and has dim * 2 bytes per row of HBM to spare should switch the image on. */
    uint64_t n_batches_without_image;
} rlr_profile;
/* enable != 0: record HIP events around each stage on the stream it is launched on
 * (adds one event pair per stage).  Disabled by default. */
int32_t rlr_profile_enable(rlr_index *idx, int32_t enable);
int32_t rlr_profile_read(rlr_index *idx, rlr_profile *out, int32_t reset);
/* Measured-peak denominators for the roofline ("re-measure on the box"), over the index's own rows
 * so that nothing else has to fit in HBM:
 *   mode 0  read-only stream: the scan's row stream with the arithmetic removed (non-temporal 16-byte loads summed
 *           up, one store per wave), the best of three launch shapes, mean of `reps` launches after one warm-up;
 *   mode 1  device-to-device copy (hipMemcpyAsync) of the first min(half of the rows, 4 GiB) into a scratch
 *           allocation; *gbps_out counts bytes read + bytes written, as copy bandwidths are usually quoted.
 *   mode 2 / 3  diagnostics: the scan kernel itself (zero query, scores into a scratch array) without / with the
 *           radix histogram it flushes with global atomics; *gbps_out = row bytes / time.
 * HIP events on the probe's own stream.  Benchmark support: no search may run on the index meanwhile. */
int32_t rlr_index_probe_bandwidth(rlr_index *idx, int32_t mode, uint32_t reps, double *gbps_out, double *ms_out);

#ifdef __cplusplus
}
#endif
#endif /* RLR_GPU_H */
