// kernels.h -- launch wrappers of the gfx950 kernels (definitions in *.hip).
// Host-callable; every wrapper only enqueues work on `stream`.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rlr {

// Every device allocation of the library goes through here.  RLR_POISON_ALLOC=1 (a test switch) fills each new
// buffer with 0xFF bytes -- NaN as f32 / binary16, huge as a counter -- so that a kernel reading memory nobody
// has written shows up as a wrong answer in the parity tests instead of hiding behind the zero pages a fresh
// process usually gets (tests/test_gpu_fuzz.py::test_poisoned_allocations).
hipError_t dev_malloc(void **p, size_t bytes);
bool poison_mode(); // RLR_POISON_ALLOC=1

// index.hip: sets the thread-local rlr_last_error() message, returns `code`
int32_t set_error(int32_t code, const char *fmt, ...);

// ---- scan.hip : wavefront-order candidate scan (HBM-bound) ----------------
struct ScanArgs {
    const void *rows;     // n_rows x pitch16 x 16 B, row-major
    const float *query;   // device, q_pitch floats (zero padded to the row pitch)
    float *scores;        // n_rows
    uint32_t *hist;       // kHistBins bins of score_key >> 21, pre-zeroed; may be null
    uint32_t n_rows;
    uint32_t dim;         // logical elements per row
    uint32_t pitch16;     // row pitch in 16-byte units
    int dtype;            // RLR_F32 / RLR_F16
    int n_cu;
    int variant;          // tuning knob, 0 = default
    // Optional: the same (zero padded) query in HOST memory.  When set and the shape allows it (f32 rows of <= 768
    // elements on the fixed kernel), the query travels in the kernel's ARGUMENTS and workgroup 0 writes it to `query`
    // for the kernels behind the scan: no upload in front of the scan (a 2.5 us kernel plus the ~4 us the host needs to
    // submit the next launch, during which the device idles).  launch_scan_takes_host_query() says whether it will.
    const float *query_host = nullptr;
};
hipError_t launch_scan(const ScanArgs &a, hipStream_t stream);
bool launch_scan_takes_host_query(const ScanArgs &a);
// one pass over the rows for 2..8 queries (a.query = n_queries x q_pitch floats, a.scores = n_queries x score_stride);
// false when the shape is not served (then nothing was launched)
bool launch_scan_multi(const ScanArgs &a, uint32_t q_pitch, uint32_t n_queries, size_t score_stride, hipStream_t s,
                       hipError_t *err);

// read-only streaming probe over `bytes` of device memory (sink: blocks * 256 floats, blocks <= n_cu * 8); shape 0..2
hipError_t launch_probe_read(const void *p, size_t bytes, float *sink, int n_cu, int shape, hipStream_t s);

// ---- select.hip : radix select / collect / sort -------------------------
// Selection state kept on the device between the stages of one query.
struct SelectState {
    uint32_t k;          // in : rank wanted (1-based), <= n
    uint32_t bin1;       // out of find1: digit of the k-th key, bits 31..21
    uint32_t k2;         // rank inside bin1
    uint32_t bin2;       // out of find2: bits 20..10
    uint32_t key_lo;     // collect threshold (guard band applied)
    uint32_t n_cand;     // collect counter (may exceed cap)
    uint32_t cap;        // in : candidate buffer capacity
    uint32_t pad;
    // fused tail (tail.hip); all four are zero between two queries of a context
    uint32_t n_work;     // candidate slots handed out so far (becomes n_cand when the tail finishes)
    uint32_t done;       // workgroups of the refine-mode stage 2 that have finished their slice
    uint32_t flags;      // bit 0: a workgroup found more candidates in its slice than its local list holds
    uint32_t mode;       // written by stage 1: 1 = direct (candidates collected and re-scored there), 2 = refine
};
hipError_t launch_collect(const float *scores, uint32_t n, SelectState *st, uint32_t *cand,
                          int n_cu, hipStream_t s);
// single-query pipeline: bin searches folded into the kernels that need them
hipError_t launch_hist2_find1(const float *scores, uint32_t n, const uint32_t *hist1, uint32_t *hist2,
                              SelectState *st, uint32_t k, uint32_t cap, int n_cu, hipStream_t s);
hipError_t launch_collect_find2(const float *scores, uint32_t n, const uint32_t *hist2, SelectState *st,
                                float two_eps, uint32_t *cand, int n_cu, hipStream_t s);
// sort `n_pad` (power of two) packed u64 descending in place; entries >= n are 0.
hipError_t launch_sort_desc(uint64_t *packed, uint32_t n_pad, hipStream_t s);

hipError_t launch_batch_select(const float *scores, uint32_t n, size_t score_stride, uint32_t q_count,
                               uint32_t *hist, SelectState *st, float two_eps, float *tau_out, uint64_t *cand,
                               uint32_t cand_stride, int n_cu, hipStream_t s);

// ---- tail.hip : everything behind the scan of a single-query search in two launches ----------------
// stage 1: bin search over hist1 (folded in), then either DIRECT -- at most `direct_max` scores sit in or above the
//          k-th score's digit-1 bin: every workgroup appends the rows of its slice at or above (bin floor - band) and
//          re-scores them in reference order on the spot (packed[slot] = (exact score, row)) -- or REFINE: the digit-2
//          histogram of that bin (what launch_hist2_find1 does);
// stage 2: DIRECT: workgroup 0 sorts and emits; REFINE: digit-2 bin search, collect + re-score per workgroup, the
//          workgroup that finishes last sorts and emits.  Both clear the two histograms for the next query.
// out == nullptr: no sort / emit (the caller orders packed[0, st->n_cand) itself); *meta receives the candidate count.
// false: the row shape does not fit the staged re-score (the caller takes the split five-launch pipeline).
// The MMR pool of a diversified search built by the workgroup that finishes the tail (pool_prepare.h) instead of the
// sort / emit: the `fetch` best candidates -> combined scores, order, cut to `need`.
struct PoolArgs {
    uint32_t fetch, need, n_rows;
    float w_e, w_l;
    uint32_t *list; // pool slot -> row
    float *comb;    // combined score per slot
    float *cosv;    // cosine per slot
    uint32_t *info; // [0] pool size, [1] status
};
struct TailArgs {
    const float *scores;
    uint32_t n;
    uint32_t *hist; // 2 * kHistBins (digit 1, digit 2), zero on entry, zero again when stage 2 has run
    SelectState *st;
    uint32_t k, cap;
    float two_eps;
    const void *rows;
    uint32_t pitch16, dim;
    int dtype;
    const float *query;
    uint64_t *packed; // cap entries
    uint64_t *out;
    uint64_t *meta;
    bool unordered;
    uint32_t direct_max;
    int n_cu;
    const PoolArgs *pool = nullptr; // non-null (with out == nullptr): the finish builds the MMR pool from the candidates
};
bool tail_fits(uint32_t pitch16, uint32_t dim, int dtype);
hipError_t launch_tail_stage1(const TailArgs &a, hipStream_t s);
hipError_t launch_tail_stage2(const TailArgs &a, hipStream_t s);

// ---- gemm.hip : batched queries, MFMA nomination + per-query exact finish ----------------
float nomination_eps(uint32_t dim, int dtype);
uint32_t batch_finish_capacity();
hipError_t launch_prep_queries(const float *q, uint32_t n_queries, uint32_t q_pitch, uint32_t dim, int dtype,
                               void *qfrag, hipStream_t s);
// scores != null: materialise nominated scores of rows [row_begin,row_end) (column = row - row_begin);
// scores == null: filter mode, append (score,row) >= tau[q] to cand[q].
hipError_t launch_gemm_nominate(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, uint32_t row_begin,
                                uint32_t row_end, const void *qfrag, uint32_t n_queries, const float *tau,
                                uint64_t *cand, uint32_t cand_stride, SelectState *st, float *scores,
                                size_t score_stride, const void *image, hipStream_t s, uint32_t *sync_ws = nullptr);
// optional binary16 nomination image of the corpus in GEMM-fragment order (see gemm.hip)
size_t image_bytes(uint32_t dim, uint64_t n_rows);
// whether launch_gemm_nominate serves rows of this width from the image (else it reads the row-major matrix)
bool gemm_image_usable(uint32_t dim);
// single-query nomination scan over the image (binary16, half the bytes of the f32 rows); dim % 64 == 0,
// query = dim floats on the device; same outputs as launch_scan; error bound nomination_eps(dim, dtype)
hipError_t launch_scan_image(const void *image, uint32_t n_rows, uint32_t dim, const float *query, float *scores,
                             uint32_t *hist, int n_cu, hipStream_t s);
hipError_t launch_build_image(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, uint32_t n_rows,
                              uint32_t tile_begin, uint32_t tile_end, void *image, hipStream_t s);
// batched finish = band cut (sort the nominated candidates, keep the guard band) -> staged reference-order
// re-score of every band (exact.hip) -> final order + emit.  `cand` is overwritten; st[q].pad receives the
// band length.
hipError_t launch_batch_finish(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, const float *queries,
                               uint32_t q_pitch, uint32_t n_queries, uint64_t *cand, uint32_t cand_stride,
                               SelectState *st, uint32_t k, float two_eps, uint64_t *out, uint32_t *status,
                               hipStream_t s);
// per query: order the candidates collected so far, move the threshold up to (k-th nominated score - band),
// drop what falls below it (st[q].n_cand, tau[q] updated); queries with a short or overflowed list are left alone
hipError_t launch_batch_tighten(uint64_t *cand, uint32_t cand_stride, SelectState *st, uint32_t n_queries, uint32_t k,
                                float two_eps, float *tau, hipStream_t s);
bool batch_rescore_fits(uint32_t pitch16, uint32_t dim, int dtype);
bool launch_batch_rescore(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, const float *queries,
                          uint32_t q_pitch, uint32_t n_queries, uint64_t *band, uint32_t band_stride,
                          const SelectState *st, hipStream_t s, hipError_t *err);

// ---- exact.hip : reference-order arithmetic --------------------------------
// packed_out[i] = pack(dot_ref(query, row[cand[i]]), cand[i]) for i < min(n_cand, cap);
// entries up to n_pad are zero-filled.  n_cand is read from the device (st->n_cand).
hipError_t launch_rescore(const void *rows, uint32_t pitch16, uint32_t dim, int dtype,
                          const float *query, const uint32_t *cand, const SelectState *st,
                          uint64_t *packed_out, uint32_t n_pad, hipStream_t s);
// LDS-staged variant for the fast path (no zero fill; also clears 2*kHistBins words at hist_clear).
bool launch_rescore_staged(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, const float *query,
                           const uint32_t *cand, const SelectState *st, uint64_t *packed_out, uint32_t n_max,
                           uint32_t *hist_clear, hipStream_t s, hipError_t *err);
// cos_out[i] = dot_ref(query, row[list[i]]) for an explicit row list.
// n_dev != null: the list length is min(*n_dev, n) (known only on the device); rows >= n_rows_clamp (markers) score row 0
hipError_t launch_score_rows(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, const float *query,
                             const uint32_t *list, uint32_t n, float *cos_out, hipStream_t s,
                             const uint32_t *n_dev = nullptr, uint32_t n_rows_clamp = 0xFFFFFFFFu);
// Reference normalize() of n rows of f32 staging data (in place), then store as dtype.
hipError_t launch_normalize_store(float *staging, uint32_t n, uint32_t dim, int do_normalize,
                                  void *rows_out, uint32_t pitch16, int dtype, float *norm_tmp,
                                  hipStream_t s);
hipError_t launch_synth(void *rows_out, uint32_t pitch16, uint32_t dim, int dtype, uint64_t row0,
                        uint32_t n, uint64_t seed, uint32_t n_clusters, float *norm_tmp,
                        hipStream_t s);
// out[i][0..dim) = widen(row[list[i]]), dense f32 with pitch `dim`.
hipError_t launch_gather_f32(const void *rows, uint32_t pitch16, uint32_t dim, int dtype,
                             const uint32_t *list, uint32_t n, float *out, hipStream_t s);
// stable compaction: dst row i = src row keep[i]
hipError_t launch_compact_rows(const void *src, void *dst, uint32_t pitch16, const uint32_t *keep,
                               uint32_t n_keep, hipStream_t s);
// gram[i*P + j] = dot_ref(pool[i], pool[j]) over a dense f32 P x dim pool.
hipError_t launch_gram(const float *pool, uint32_t P, uint32_t dim, float *gram, uint32_t n_queries, hipStream_t s);
// the same with the pool rows read straight from the index through list[q * P + i] (no gathered f32 copy)
hipError_t launch_gram_rows(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, const uint32_t *list, uint32_t P,
                            float *gram, uint32_t n_queries, hipStream_t s);
// Optional tail of the greedy kernel for a single pool (the fused search paths): the picks go straight to (pinned)
// memory as [row | cos | combined | lexical] x k_cap, then n, status -- no separate emit launch.
struct MmrEmit {
    const uint32_t *list = nullptr; // pool slot -> row
    const float *comb = nullptr, *cosv = nullptr, *lexv = nullptr; // per pool slot (lexv may be null: zeros)
    const uint32_t *info = nullptr; // info[1] != 0: the pool is not usable, n = 0
    uint32_t k_cap = 0;
    uint32_t *h_out = nullptr;      // null: no emit
};
// greedy MMR over the gram matrix; out_order/out_mmr/out_n on the device.
hipError_t launch_mmr_greedy(const float *gram, const float *scores, uint32_t P, uint32_t k,
                             float lambda, uint32_t *out_order, float *out_mmr, uint32_t *out_n,
                             const uint32_t *sizes, uint32_t n_queries, hipStream_t s, const MmrEmit *emit = nullptr);

// ---- q8.hip : optional 8-bit nomination copy of f32 rows (single-query scans at a quarter of the bytes) ----
hipError_t launch_q8_build(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, uint32_t row_begin, uint32_t n_rows,
                           void *q8, float *scale, uint32_t *stats, hipStream_t s);
hipError_t launch_q8_scan(const void *q8, const float *scale, uint32_t n_rows, uint32_t dim, const float *query,
                          float *scores, uint32_t *hist, int n_cu, hipStream_t s);
float q8_arith_eps(uint32_t dim, float scale_max, float q_norm);

} // namespace rlr
