/* rlr_lexical.h -- C ABI of the GPU-resident BM25 term of the hybrid score
 * (SURVEY.md section 8(f) row f3).
 *
 * Replaces, in /root/reference/src/rag_engine.rs:
 *   struct LexicalIndex                      :2083-2090
 *   LexicalIndex::add_chunk / remove_chunk   :2106-2167
 *   LexicalIndex::score  (BM25, k1 1.5, b 0.75, idf floored at 0, sort desc, truncate)  :2169-2225
 *   LexicalIndex::contains / clear           :2227-2229, :2098-2104
 * and is called from RagEngine::search at :505 (`lexical_index.score(query, 5 * top_k)`).
 *
 * Split of the work
 *   host language   `tokenize` (:2242-2247) -- Unicode `char::is_alphanumeric` splitting, the
 *                   >= 3 BYTE filter and `to_lowercase` stay where the Unicode tables are (Rust
 *                   std; Python's str methods in the veneer).  Tokens cross the boundary as one
 *                   UTF-8 string, single-space separated, in text order.
 *   this library    term dictionary, per-row term counts (host), postings in HBM (CSR: rows +
 *                   term frequencies per term, document lengths), the BM25 arithmetic and the
 *                   top-`limit` selection on the GPU.
 *
 * Rows are the rows of the rlr_index the chunks' embeddings live in: add_chunk(row) follows
 * rlr_index_append, rlr_lexical_remove_rows follows rlr_index_delete_rows with the same stable
 * compaction (surviving rows keep their order and move down).
 *
 * Arithmetic: IEEE binary32, the reference's expression order (no FMA):
 *   avg   = total_length as f32 / total_docs as f32
 *   idf   = max(ln((N - df + 0.5) / (df + 0.5)), 0)          (host, logf of the C library)
 *   denom = tf + 1.5 * ((1 - 0.75) + 0.75 * (dl / avg))
 *   score = (idf * (tf * 2.5)) / denom,  accumulated per document over the query's unique terms.
 * Where the reference is unspecified this library defines:
 *   - terms are accumulated in order of first occurrence in the query (the reference iterates a
 *     HashSet, so its f32 sum order -- and its last bit -- varies from run to run);
 *   - equal scores order by lower row first (the reference's order among ties is arbitrary);
 *   - documents whose total is exactly 0.0 (every matching term has idf 0) are not returned:
 *     in the caller (:515-532) they contribute lexical = 0 exactly like absent documents.
 * Threading: the handle carries a readers/writer lock like the reference's RwLock around its engine: score calls
 * from several threads run concurrently, each on its own stream and workspace (up to 8; further callers wait for
 * one); add / remove / clear take the lock exclusively and wait for running score calls.  The first score call
 * after a mutation rebuilds the device postings under the exclusive lock.  destroy needs every call returned.
 */
#ifndef RLR_LEXICAL_H
#define RLR_LEXICAL_H

#include "rlr_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rlr_lexical rlr_lexical;

#define RLR_LEXICAL_MAX_LIMIT 8192u /* most (row, score) pairs one score call returns */

/* LexicalIndex::new (:2093-2095) on HIP device `device_id`. */
int32_t rlr_lexical_create(int32_t device_id, rlr_lexical **out);
void rlr_lexical_destroy(rlr_lexical *lex);

/* LexicalIndex::add_chunk (:2106-2137).  `tokens`: output of the host's tokenize(), space
 * separated, `len` bytes.  A row that is already present is replaced (:2107-2109); a chunk
 * without tokens is not indexed (:2112-2114). */
int32_t rlr_lexical_add_chunk(rlr_lexical *lex, uint64_t row, const char *tokens, size_t len);

/* LexicalIndex::remove_chunk (:2139-2167) for a set of rows + the row compaction that
 * rlr_index_delete_rows performs.  Rows may be given in any order; unknown rows are ignored.  The device postings
 * follow by an ordered compaction of both segments (three launches per segment, survivors renumbered in place of a
 * host rebuild and re-upload: 3 ms instead of 0.17 s at 26.6 M postings). */
int32_t rlr_lexical_remove_rows(rlr_lexical *lex, const uint64_t *rows, uint32_t n);

/* LexicalIndex::clear (:2098-2104). */
int32_t rlr_lexical_clear(rlr_lexical *lex);

/* LexicalIndex::contains (:2227-2229): 1 / 0, negative on error. */
int32_t rlr_lexical_contains(rlr_lexical *lex, uint64_t row);

/* total_docs, total_length (:2088-2089), number of distinct live terms and of postings. */
int32_t rlr_lexical_info(rlr_lexical *lex, uint64_t *total_docs, uint64_t *total_length,
                         uint64_t *n_terms, uint64_t *n_postings);

/* How the postings sit on the device (no reference counterpart: the reference's HashMap needs no rebuild).  The
 * main segment holds the rows present at the last full rebuild, the appended segment the rows added since; a commit
 * that finds only appends rebuilds the appended segment alone.  select_retries: score calls whose sampled top-`limit`
 * selection handed the query back to the exact radix passes (expected: fewer than one in 10^5).  Any pointer may be
 * null. */
int32_t rlr_lexical_segments(rlr_lexical *lex, uint64_t *main_postings, uint64_t *appended_postings,
                             uint64_t *full_rebuilds, uint64_t *append_rebuilds, uint64_t *select_retries);

/* LexicalIndex::score (:2169-2225).  `query_tokens` like `tokens` above.  Writes at most
 * min(limit, RLR_LEXICAL_MAX_LIMIT) pairs, ordered (score desc, row asc); limit == 0 means
 * "no truncation" in the reference (:2220) and is served up to RLR_LEXICAL_MAX_LIMIT pairs.
 * The first call after a mutation brings the device postings up to date: appends (rows beyond every row present at
 * the last full rebuild) cost O(terms + appended postings); a replaced row of the main segment, or an appended segment
 * that has outgrown an eighth of the main one, costs a full rebuild (removals are applied to the device postings by
 * rlr_lexical_remove_rows itself). */
int32_t rlr_lexical_score(rlr_lexical *lex, const char *query_tokens, size_t len, uint32_t limit,
                          uint64_t *rows_out, float *scores_out, uint32_t *n_out);

/* Convenience tokenizer for hosts without Unicode tables: exact for ASCII text; every
 * non-ASCII code point is treated as alphanumeric and left unchanged (the reference would
 * split at non-ASCII punctuation and lower-case non-ASCII letters).  Writes the space-joined
 * tokens to `out` (capacity `cap`), *out_len = bytes needed; RLR_E_RANGE if cap is too small. */
int32_t rlr_tokenize_ascii(const char *text, size_t len, char *out, size_t cap, size_t *out_len);

#ifdef __cplusplus
}
#endif
#endif /* RLR_LEXICAL_H */
