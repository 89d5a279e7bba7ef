// lexical_internal.h -- library-internal view of a BM25 scoring call that leaves its result on the device, so the hybrid
// search (csrc/engine.cpp: rlr_engine_search_text) can run it beside the cosine scan and blend without a host round trip.
#pragma once

#include <stddef.h>
#include <stdint.h>

struct rlr_lexical;
struct rlr_index;

namespace rlr {

// An enqueued scoring call.  From lexical_enqueue to lexical_finish the caller holds one of the index' workspaces and
// its readers' lock: mutators wait, other scoring calls do not.
struct LexPending {
    rlr_lexical *lx = nullptr;
    void *ws = nullptr;                 // the leased workspace
    void *stream = nullptr;             // hipStream_t: its stream; `ready` is recorded there behind the last kernel
    void *ready = nullptr;              // hipEvent_t (opaque here: csrc/engine.cpp is built without the HIP headers)
    const uint64_t *d_packed = nullptr; // pack_result(score, row) keys, the best `*d_count` by (score desc, row asc)
    const uint32_t *d_count = nullptr;  // how many of them are valid (<= limit)
    uint32_t limit = 0;                 // 0: no document can match (empty index / unknown terms) -- nothing was enqueued
    bool locked = false;
    bool may_retry = false; // the sampled selection was used: the count may read kLexicalRetry
    bool unpacked = false;  // the scoring stream also wrote the rows / scores / header a LexSink asked for (in front of `ready`)
};

// Where a hybrid search wants a scoring call's result unpacked (index.hip: lex_unpack_kernel): lexical_enqueue launches that
// on ITS stream in front of `ready`, so the search's own stream finds rows, scores and header (count, max score) in place
// when it has joined -- one dependent launch less behind the join.
struct LexSink {
    uint32_t *d_rows = nullptr; // `n_bound` slots
    float *d_scores = nullptr;  // `n_bound` slots
    void *d_header = nullptr;   // HybridLexHeader
    uint32_t n_bound = 0;       // most pairs the search takes
    uint32_t n_index_rows = 0;  // rows of the embedding index (pairs beyond it are marked, see lex_unpack_kernel)
};
// (index.hip) no error check inside: the caller's hipGetLastError sees it
void launch_lex_unpack(const uint64_t *d_packed, const uint32_t *d_count, uint32_t limit, const LexSink &sink, void *stream);

// LexicalIndex::score (rag_engine.rs:2169-2225) up to the ordered result list in device memory.  No synchronisation.
// need_sorted = false: d_packed holds the same set in no particular order (saves the final LDS sort; lexical_fetch
// needs the sorted form)
// exact_passes = false lets large candidate sets go through the sampled 3-launch selection: *d_count (and lexical_fetch's
// *n_out) may then read kLexicalRetry -- repeat the call with exact_passes = true
int32_t lexical_enqueue(rlr_lexical *lx, const char *tokens, size_t len, uint32_t limit, LexPending *out, bool need_sorted,
                        bool exact_passes, const LexSink *sink = nullptr);
constexpr uint32_t kLexicalRetry = 0xFFFFFFFFu;
// copy the result to the host (synchronises the scoring stream); valid between enqueue and finish
int32_t lexical_fetch(LexPending *p, uint64_t *rows_out, float *scores_out, uint32_t *n_out);
// rlr_lexical_score without the sampled attempt, for a query whose fused search came back with kLexicalRetry (status 3):
// counts the retry (rlr_lexical_segments: select_retries) and goes straight to the exact radix passes
int32_t lexical_score_exact(rlr_lexical *lx, const char *tokens, size_t len, uint32_t limit, uint64_t *rows_out,
                            float *scores_out, uint32_t *n_out);
// hand the workspace back once every consumer of d_packed has finished (the caller synchronised them);
// ok = false: something failed after the enqueue -- the workspace is re-zeroed before its next use
void lexical_finish(LexPending *p, bool ok);

// index.hip: rlr_search_hybrid (include/rlr_gpu.h) in two enqueues, so that the BM25 kernels can be launched in between:
// begin puts the scan .. sort of the need + n_lex_bound + 8 best rows by cosine on the index' stream (*fallback != 0: not
// covered by the fused kernels, no ticket); finish joins `lex` (null or limit 0: no lexical pair) by event, enqueues
// blend .. results, synchronises and consumes the ticket; abort drains and frees a ticket that will not be finished.
struct HybridTicket;
// behind_scan (may be null): called once the scan is queued and before the select .. sort launches are -- the place to launch
// work for another stream that should run beside the scan (a BM25 call hands the LexSink it gets on to lexical_enqueue);
// not called when *fallback != 0; its error is begin's error.
int32_t search_hybrid_begin(rlr_index *ix, const float *query, uint32_t need, uint32_t k, float lambda, int32_t diversify,
                            float w_embedding, float w_lexical, uint32_t n_lex_bound, float guard_eps, HybridTicket **ticket,
                            int32_t *fallback, int32_t (*behind_scan)(void *, const LexSink *) = nullptr,
                            void *behind_scan_arg = nullptr);
int32_t search_hybrid_finish(HybridTicket *ticket, const LexPending *lex, uint64_t *rows_out, float *cos_out, float *score_out,
                             float *lex_out, uint32_t *n_out, int32_t *fallback);
void search_hybrid_abort(HybridTicket *ticket);

} // namespace rlr
