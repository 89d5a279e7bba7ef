// lexical.hip -- the C ABI of include/rlr_lexical.h: BM25 term of the hybrid score with the
// postings resident in HBM (reference LexicalIndex,
// /root/reference/src/rag_engine.rs:2083-2237).
//
// Host side: term dictionary, per-row term counts, corpus statistics; CSR postings are rebuilt
// and uploaded lazily by the first score call after a mutation.
// Device side, per query (the call's own workspace and stream out of a small pool, so calls on different host threads
// overlap; no host round trip until the results are read):
//   bm25_terms_lds_kernel  all unique query terms in one launch, the ROWS partitioned over the workgroups: every
//                      posting adds its BM25 contribution to its row's accumulator in query-term order (the f32 sum
//                      order of the reference -- deterministic), the accumulators of a workgroup's rows in LDS; rows
//                      whose sum becomes positive are appended to a compact "touched" list.  (bm25_terms_kernel: the
//                      same with the accumulators in device memory, for indexes beyond a few million rows;
//                      bm25_term_kernel: one launch per term, an A/B switch);
//   <= 8192 postings   lex_sort_kernel: packed (score, row) keys sorted in LDS;
//   more, limit <= 4096  sampled selection, 3 launches: lex_sample_kernel (threshold key from a strided sample),
//                      lex_filter_kernel (one pass, ~1.5 limit candidates), lex_final_kernel (exact limit-th key
//                      among them by an LDS radix select); hands the query to the exact path when the candidate
//                      list overflows or comes out short (< 1e-5 per query);
//   otherwise / retry  8 x lex_select_pass_kernel (MSD radix select, 8-bit digits over the unique 64-bit keys --
//                      exact under massive score ties, which BM25 produces whenever tf and document length
//                      repeat; the first pass packs the keys) + lex_collect_kernel + lex_sort_kernel;
//   lex_clear_kernel   restores the all-zero accumulator by visiting only the touched rows, and zeroes the control block
//                      (counters, histograms) the workspace's NEXT call works in -- there are two, used in turn.
// All of it is integer/f32 work bounded by HBM latency, not bandwidth: a query touches
// sum(df) postings x 8 B.
#include "../../include/rlr_lexical.h"
#include "common.h"
#include "kernels.h"
#include "lds_select.h"
#include "lexical_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

using namespace rlr;

namespace {

#define LEX_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return set_error(e_ == hipErrorOutOfMemory ? RLR_E_OOM : RLR_E_HIP, "%s failed: %s (%s:%d)", \
                             #call, hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)
#define LEX_TRY(call)          \
    do {                       \
        int32_t s_ = (call);   \
        if (s_ != RLR_OK)      \
            return s_;         \
    } while (0)

constexpr uint32_t kMaxLimit = RLR_LEXICAL_MAX_LIMIT; // LDS sort capacity (64 KB of u64 keys)
constexpr int kPasses = 8;                            // 8-bit digits over 64-bit keys
constexpr float kK1 = 1.5f, kB = 0.75f;               // rag_engine.rs:2191-2192

// control block in device memory, all zero between queries
struct LexControl {
    uint32_t n_touched;
    uint32_t n_sel;
    uint32_t n_cand;  // sampled selection: candidates at or above `thr`
    uint32_t pad;
    uint64_t prefix[kPasses + 1]; // prefix[p]: the top 8*p key bits of the k-th largest key
    uint32_t k_rem[kPasses + 1];  // rank still wanted inside that prefix (1-based)
    uint32_t pad2;
    uint64_t thr;     // sampled selection: the (score, row) key every candidate reaches
    uint32_t hist[kPasses][256];
};

// ---------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------
// wave-aggregated append: lanes with `flag` get consecutive slots of list[]
__device__ inline uint32_t wave_append_slot(bool flag, uint32_t *counter)
{
    const uint64_t mask = __ballot(flag);
    if (mask == 0)
        return 0;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll(static_cast<long long>(mask)) - 1;
    uint32_t base = 0;
    if (lane == leader)
        base = atomicAdd(counter, static_cast<uint32_t>(__popcll(mask)));
    base = __shfl(base, leader);
    return base + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
}

// LexicalIndex::score inner loop, rag_engine.rs:2200-2213, for the postings of one term
__global__ __launch_bounds__(256) void bm25_term_kernel(const uint32_t *__restrict__ post_row,
                                                        const uint32_t *__restrict__ post_tf, uint32_t cnt,
                                                        const uint32_t *__restrict__ doc_len, float avg, float idf,
                                                        float *__restrict__ scores, uint32_t *__restrict__ touched,
                                                        LexControl *__restrict__ ctl)
{
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t i0 = blockIdx.x * 256; i0 < cnt; i0 += stride) {
        const uint32_t i = i0 + threadIdx.x;
        bool first_touch = false;
        uint32_t row = 0;
        if (i < cnt) {
            row = post_row[i];
            const float dl = static_cast<float>(doc_len[row]);
            const float tf = static_cast<float>(post_tf[i]);
            const float denom = tf + kK1 * ((1.0f - kB) + kB * (dl / avg));
            if (dl != 0.0f && denom != 0.0f) {
                const float sc = idf * (tf * (kK1 + 1.0f)) / denom;
                const float old = scores[row];
                const float now = old + sc; // `*scores.entry(doc).or_insert(0.0) += score`
                scores[row] = now;
                first_touch = old == 0.0f && now > 0.0f;
            }
        }
        const uint32_t slot = wave_append_slot(first_touch, &ctl->n_touched);
        if (first_touch)
            touched[slot] = row;
    }
}

// bitonic sort (descending) of n_pad keys in LDS, n_pad a power of two <= kMaxLimit
__device__ inline void lds_sort_desc(uint64_t *s, uint32_t n_pad)
{
    for (uint32_t kk = 2; kk <= n_pad; kk <<= 1) {
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n_pad; i += blockDim.x) {
                const uint32_t ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = s[i], b = s[ixj];
                    const bool desc = (i & kk) == 0;
                    if (desc ? (a < b) : (a > b)) {
                        s[i] = b;
                        s[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// All the query's terms in ONE launch (it was one launch per term and segment: ~6 x 5 us of dependent launches and as many
// host-side launch calls on the critical path of a hybrid search).  The f32 sum order per row must stay the term order
// (`*scores.entry(doc) += score` in query-term order, :2195-2219), so the ROWS are partitioned instead of the terms: workgroup w
// owns the rows [w n / G, (w + 1) n / G); every posting list is sorted by row, so the workgroup's share of each list is one
// contiguous range found by a binary search (all terms and both segments at once, one lane each), and it walks the terms in
// order with a workgroup barrier in between -- no row is ever touched by two workgroups, no grid-wide synchronisation.
constexpr int kTermsPerLaunch = 16;
struct TermBatch {
    uint32_t n_terms;
    uint32_t cnt_m[kTermsPerLaunch], cnt_d[kTermsPerLaunch];
    uint64_t off_m[kTermsPerLaunch], off_d[kTermsPerLaunch];
    float idf[kTermsPerLaunch];
};

__device__ inline uint32_t lower_bound_rows(const uint32_t *__restrict__ rows, uint32_t cnt, uint32_t key)
{
    uint32_t lo = 0, hi = cnt; // first index with rows[i] >= key
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (rows[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void bm25_terms_kernel(TermBatch tb, const uint32_t *__restrict__ post_row,
                                                         const uint32_t *__restrict__ post_tf, const uint32_t *__restrict__ dpost_row,
                                                         const uint32_t *__restrict__ dpost_tf, const uint32_t *__restrict__ doc_len,
                                                         uint32_t n_rows, float avg, float *__restrict__ scores,
                                                         uint32_t *__restrict__ touched, LexControl *__restrict__ ctl)
{
    __shared__ uint32_t s_range[kTermsPerLaunch][4]; // [term][main lo, main hi, appended lo, appended hi]
    const uint32_t r0 = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x) * n_rows / gridDim.x);
    const uint32_t r1 = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x + 1) * n_rows / gridDim.x);
    if (threadIdx.x < 4 * kTermsPerLaunch) {
        const uint32_t t = threadIdx.x >> 2, which = threadIdx.x & 3;
        uint32_t v = 0;
        if (t < tb.n_terms) {
            const bool delta = which >= 2;
            const uint32_t cnt = delta ? tb.cnt_d[t] : tb.cnt_m[t];
            if (cnt)
                v = lower_bound_rows((delta ? dpost_row + tb.off_d[t] : post_row + tb.off_m[t]), cnt, (which & 1) ? r1 : r0);
        }
        s_range[t][which] = v;
    }
    __syncthreads();
    for (uint32_t t = 0; t < tb.n_terms; ++t) {
        const float idf = tb.idf[t];
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) { // a row lives in exactly one segment: no barrier between the two
            const uint32_t lo = s_range[t][2 * seg], hi = s_range[t][2 * seg + 1];
            const uint32_t *rows = seg ? dpost_row + tb.off_d[t] : post_row + tb.off_m[t];
            const uint32_t *tfs = seg ? dpost_tf + tb.off_d[t] : post_tf + tb.off_m[t];
            for (uint32_t i0 = lo; i0 < hi; i0 += 256) { // (uniform bounds: every wave runs the same trips)
                const uint32_t i = i0 + threadIdx.x;
                bool first_touch = false;
                uint32_t row = 0;
                if (i < hi) {
                    row = rows[i];
                    const float dl = static_cast<float>(doc_len[row]);
                    const float tf = static_cast<float>(tfs[i]);
                    const float denom = tf + kK1 * ((1.0f - kB) + kB * (dl / avg));
                    if (dl != 0.0f && denom != 0.0f) {
                        const float sc = idf * (tf * (kK1 + 1.0f)) / denom;
                        const float old = scores[row];
                        const float now = old + sc; // `*scores.entry(doc).or_insert(0.0) += score`
                        scores[row] = now;
                        first_touch = old == 0.0f && now > 0.0f;
                    }
                }
                const uint32_t slot = wave_append_slot(first_touch, &ctl->n_touched);
                if (first_touch)
                    touched[slot] = row;
            }
        }
        __syncthreads(); // the next term adds to the sums this one wrote (other threads' rows included)
    }
}

// The same launch with the accumulators of the workgroup's rows in LDS (<= kLdsRows of them: indexes up to a few million
// rows).  The global form above walks a chain of dependent memory round trips per term -- posting -> document length ->
// accumulator -> first-touch slot (an atomic on one counter per wave) -> store -- behind seventeen more for the range
// searches: ~40 us for six terms over 100 k rows, the longest kernel beside the scan of a text search.  Here the document
// lengths and accumulators of the rows are loaded once (coalesced, in flight during the searches), the searches take one
// round as a rule, a term's postings are on their way while the previous term is added, the sums stay in LDS between terms, and the
// touched rows are appended with one reservation per workgroup.  The f32 sum order per row is the term order, as before.
constexpr uint32_t kLdsRows = 2048;
constexpr int kSearchPerWave = 4;        // range searches a wave runs at a time
constexpr uint32_t kSearchWin = 512;     // entries of the window around a search's estimated place

// First index with rows[i] >= key, by one wavefront: 64 probes per round (the last entry of each 64th of the range), three
// dependent loads for a 100 k-entry list.  Uniform result.
__device__ inline uint32_t wave_lower_bound_rows(const uint32_t *__restrict__ rows, uint32_t cnt, uint32_t key, uint32_t lane)
{
    uint32_t lo = 0, hi = cnt;
    while (hi - lo > 64) {
        const uint32_t step = (hi - lo + 63) / 64;
        const uint32_t begin = min(lo + lane * step, hi), end = min(lo + (lane + 1) * step, hi);
        const bool below = begin < end && rows[end - 1] < key; // this lane's whole block lies below the key
        const uint32_t c = static_cast<uint32_t>(__popcll(__ballot(below)));
        const uint32_t nlo = min(lo + c * step, hi);
        hi = min(lo + (c + 1) * step, hi);
        lo = nlo;
    }
    const uint32_t i = lo + lane;
    const bool below = i < hi && rows[i] < key;
    return lo + static_cast<uint32_t>(__popcll(__ballot(below)));
}
__global__ __launch_bounds__(256) void bm25_terms_lds_kernel(TermBatch tb, const uint32_t *__restrict__ post_row,
                                                             const uint32_t *__restrict__ post_tf, const uint32_t *__restrict__ dpost_row,
                                                             const uint32_t *__restrict__ dpost_tf, const uint32_t *__restrict__ doc_len,
                                                             uint32_t n_rows, float avg, float *__restrict__ scores,
                                                             uint32_t *__restrict__ touched, LexControl *__restrict__ ctl)
{
    __shared__ uint32_t s_range[kTermsPerLaunch][4]; // [term][main lo, main hi, appended lo, appended hi]
    __shared__ float s_sc[kLdsRows], s_dl[kLdsRows], s_nk[kLdsRows];
    __shared__ uint8_t s_fl[kLdsRows]; // bit 0: the sum became positive (first touch), bit 1: the sum was written
    __shared__ uint32_t s_cnt, s_base;
    __shared__ uint32_t s_id[kTermsPerLaunch * 4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r0 = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x) * n_rows / gridDim.x);
    const uint32_t r1 = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x + 1) * n_rows / gridDim.x);
    const uint32_t nr = r1 - r0; // <= kLdsRows (the launch sees to it)
    for (uint32_t r = tid; r < nr; r += 256) {
        s_sc[r] = scores[r0 + r]; // zero unless an earlier launch of this query (more than 16 terms) was here
        const float dl = static_cast<float>(doc_len[r0 + r]);
        s_dl[r] = dl;
        s_nk[r] = kK1 * ((1.0f - kB) + kB * (dl / avg)); // the row's share of every posting's denominator (same operations, once)
        s_fl[r] = 0;
    }
    if (tid == 0)
        s_cnt = 0;
    // This workgroup's share of every posting list (sorted by row): first index >= r0 and first index >= r1.  A lane's binary
    // search took seventeen dependent loads per list, each of them microseconds while the scan kernel next door saturates
    // the memory system; probing 64 or 512 places per round took few rounds but so many cache lines (a line per probe,
    // ~400 workgroups x 12 searches) that the traffic itself became the cost.  So: the place is ESTIMATED from the key
    // (postings spread evenly over the rows land within a few hundred entries of cnt * key / n_rows), one coalesced window of
    // 512 entries around the estimate is read, and the window decides if the key falls inside it; else the 64-ary search
    // below takes the side it points to.  A wave runs four searches at a time, their windows in flight together.
    uint32_t nz = 0;
    for (uint32_t id = 0; id < tb.n_terms * 4; ++id) {
        const uint32_t cnt = (id & 2) ? tb.cnt_d[id >> 2] : tb.cnt_m[id >> 2];
        if (cnt == 0) {
            if (tid == 0)
                s_range[id >> 2][id & 3] = 0;
            continue;
        }
        if (tid == 0)
            s_id[nz] = id;
        ++nz;
    }
    __syncthreads();
    for (uint32_t b0 = wave * kSearchPerWave; b0 < nz; b0 += 4 * kSearchPerWave) {
        uint32_t wlo[kSearchPerWave], whi[kSearchPerWave], key[kSearchPerWave], v[kSearchPerWave][kSearchWin / 64];
#pragma unroll
        for (int j = 0; j < kSearchPerWave; ++j) {
            wlo[j] = whi[j] = key[j] = 0;
            if (b0 + j >= nz)
                continue;
            const uint32_t id = s_id[b0 + j], t = id >> 2;
            const uint32_t *rows = (id & 2) ? dpost_row + tb.off_d[t] : post_row + tb.off_m[t];
            const uint32_t cnt = (id & 2) ? tb.cnt_d[t] : tb.cnt_m[t];
            key[j] = (id & 1) ? r1 : r0;
            // (an estimate: f32 is exact enough, and a 64-bit integer division is ~200 instructions, four times per wave)
            const uint32_t est = min(cnt, static_cast<uint32_t>(static_cast<float>(cnt) * (static_cast<float>(key[j]) / static_cast<float>(n_rows))));
            whi[j] = min(max(est, kSearchWin / 2) + kSearchWin / 2, cnt);
            wlo[j] = whi[j] > kSearchWin ? whi[j] - kSearchWin : 0u;
#pragma unroll
            for (uint32_t u = 0; u < kSearchWin / 64; ++u) {
                const uint32_t i = wlo[j] + lane + 64 * u;
                v[j][u] = i < whi[j] ? rows[i] : 0xFFFFFFFFu;
            }
        }
#pragma unroll
        for (int j = 0; j < kSearchPerWave; ++j) {
            if (b0 + j >= nz)
                continue;
            const uint32_t id = s_id[b0 + j], t = id >> 2;
            const uint32_t *rows = (id & 2) ? dpost_row + tb.off_d[t] : post_row + tb.off_m[t];
            const uint32_t cnt = (id & 2) ? tb.cnt_d[t] : tb.cnt_m[t];
            uint32_t c = 0; // entries of the window below the key (a prefix of it: the list is sorted)
#pragma unroll
            for (uint32_t u = 0; u < kSearchWin / 64; ++u)
                c += static_cast<uint32_t>(__popcll(__ballot(v[j][u] < key[j])));
            uint32_t res;
            if ((wlo[j] == 0 || c > 0) && (whi[j] == cnt || c < whi[j] - wlo[j]))
                res = wlo[j] + c; // the boundary lies inside the window
            else if (c == 0)
                res = wave_lower_bound_rows(rows, wlo[j], key[j], lane); // the whole window is at or above the key
            else
                res = whi[j] + wave_lower_bound_rows(rows + whi[j], cnt - whi[j], key[j], lane);
            if (lane == 0)
                s_range[t][id & 3] = res;
        }
    }
    __syncthreads();
    // chunks of <= 256 postings in (term, segment) order; a row lives in exactly one segment and once in a list, so only a
    // change of TERM needs a barrier (the next term adds to sums other threads wrote)
    auto seek = [&](uint32_t &t, uint32_t &seg, uint32_t &i0) { // the first chunk at or behind (t, seg, i0) that holds postings
        while (t < tb.n_terms && i0 >= s_range[t][2 * seg + 1]) {
            if (seg == 0) {
                seg = 1;
            } else {
                seg = 0;
                ++t;
            }
            if (t < tb.n_terms)
                i0 = s_range[t][2 * seg];
        }
    };
    auto fetch = [&](uint32_t t, uint32_t seg, uint32_t i0, uint32_t *row, uint32_t *tf) -> bool {
        if (t >= tb.n_terms)
            return false;
        const uint32_t i = i0 + tid;
        if (i >= s_range[t][2 * seg + 1])
            return false;
        *row = (seg ? dpost_row + tb.off_d[t] : post_row + tb.off_m[t])[i];
        *tf = (seg ? dpost_tf + tb.off_d[t] : post_tf + tb.off_m[t])[i];
        return true;
    };
    uint32_t t = 0, seg = 0, i0 = s_range[0][0];
    seek(t, seg, i0);
    uint32_t row = 0, tfw = 0;
    bool ok = fetch(t, seg, i0, &row, &tfw);
    while (t < tb.n_terms) {
        uint32_t t2 = t, seg2 = seg, j0 = i0 + 256;
        seek(t2, seg2, j0);
        uint32_t row2 = 0, tfw2 = 0;
        const bool ok2 = fetch(t2, seg2, j0, &row2, &tfw2); // on its way while this chunk is added
        if (ok) {
            const uint32_t r = row - r0;
            const float dl = s_dl[r];
            const float tf = static_cast<float>(tfw);
            const float denom = tf + s_nk[r];
            if (dl != 0.0f && denom != 0.0f) {
                const float sc = tb.idf[t] * (tf * (kK1 + 1.0f)) / denom;
                const float old = s_sc[r];
                const float now = old + sc; // `*scores.entry(doc).or_insert(0.0) += score`
                s_sc[r] = now;
                s_fl[r] = static_cast<uint8_t>(s_fl[r] | 2u | ((old == 0.0f && now > 0.0f) ? 1u : 0u));
            }
        }
        if (t2 != t)
            __syncthreads();
        t = t2;
        seg = seg2;
        i0 = j0;
        row = row2;
        tfw = tfw2;
        ok = ok2;
    }
    __syncthreads();
    // the sums back to the dense array; the rows that became positive into the touched list, one reservation per workgroup
    uint32_t mine = 0;
    for (uint32_t r = tid; r < nr; r += 256) {
        const uint32_t fl = s_fl[r];
        if (fl & 2u)
            scores[r0 + r] = s_sc[r];
        mine += fl & 1u;
    }
    uint32_t at = mine ? atomicAdd(&s_cnt, mine) : 0u;
    __syncthreads();
    if (tid == 0)
        s_base = s_cnt ? atomicAdd(&ctl->n_touched, s_cnt) : 0u;
    __syncthreads();
    at += s_base;
    if (mine)
        for (uint32_t r = tid; r < nr; r += 256)
            if (s_fl[r] & 1u)
                touched[at++] = r0 + r;
}

// `results.sort_by(score desc)` + `truncate(limit)` (:2218-2222) when everything fits one workgroup
template <bool FROM_TOUCHED>
__global__ __launch_bounds__(1024) void lex_sort_kernel(const float *__restrict__ scores,
                                                        const uint32_t *__restrict__ touched,
                                                        const uint64_t *__restrict__ sel,
                                                        const LexControl *__restrict__ ctl, uint32_t limit,
                                                        uint64_t *__restrict__ out_keys, uint32_t *__restrict__ out_n)
{
    __shared__ uint64_t s[kMaxLimit];
    uint32_t n = FROM_TOUCHED ? ctl->n_touched : ctl->n_sel;
    n = min(n, kMaxLimit);
    uint32_t n_pad = 1;
    while (n_pad < n)
        n_pad <<= 1;
    for (uint32_t i = threadIdx.x; i < n_pad; i += blockDim.x) {
        uint64_t v = 0;
        if (i < n) {
            if constexpr (FROM_TOUCHED) {
                const uint32_t row = touched[i];
                v = pack_result(scores[row], row);
            } else {
                v = sel[i];
            }
        }
        s[i] = v;
    }
    __syncthreads();
    lds_sort_desc(s, n_pad);
    const uint32_t m = min(n, limit);
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x)
        out_keys[i] = s[i];
    if (threadIdx.x == 0)
        *out_n = m;
}

// The (prefix, rank) the radix select has reached before pass `pass`, derived from the previous
// pass's histogram by every workgroup on its own (256 bins: one serial walk by thread 0).
__device__ inline void lex_derive_state(const LexControl *ctl, uint32_t limit, int pass, uint64_t *prefix, uint32_t *k_rem)
{
    __shared__ uint64_t s_prefix;
    __shared__ uint32_t s_k;
    if (threadIdx.x == 0) {
        if (pass == 0) {
            s_prefix = 0;
            s_k = min(limit, ctl->n_touched);
        } else {
            uint32_t k = ctl->k_rem[pass - 1];
            const uint32_t *h = ctl->hist[pass - 1];
            uint32_t d = 255, seen = 0;
            for (int b = 255; b >= 0; --b) {
                const uint32_t c = h[b];
                if (seen + c >= k) {
                    d = static_cast<uint32_t>(b);
                    break;
                }
                seen += c;
            }
            s_prefix = (ctl->prefix[pass - 1] << 8) | d;
            s_k = k - seen;
        }
    }
    __syncthreads();
    *prefix = s_prefix;
    *k_rem = s_k;
}

// PACK (pass 0 only): the keys do not exist yet -- build them from the accumulators on the way (this used to be a
// launch of its own in front of the eight passes).
template <bool PACK>
__global__ __launch_bounds__(256) void lex_select_pass_kernel(uint64_t *__restrict__ keys, LexControl *__restrict__ ctl,
                                                              uint32_t limit, int pass, const float *__restrict__ scores,
                                                              const uint32_t *__restrict__ touched)
{
    __shared__ uint32_t s_h[256];
    uint64_t prefix;
    uint32_t k_rem;
    lex_derive_state(ctl, limit, pass, &prefix, &k_rem);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->prefix[pass] = prefix;
        ctl->k_rem[pass] = k_rem;
    }
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t n = ctl->n_touched;
    const int shift = 56 - 8 * pass;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        uint64_t key;
        if constexpr (PACK) {
            const uint32_t row = touched[i];
            key = pack_result(scores[row], row);
            keys[i] = key;
        } else {
            key = keys[i];
        }
        const bool match = pass == 0 || (key >> (shift + 8)) == prefix;
        if (match)
            atomicAdd(&s_h[(key >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    const uint32_t c = s_h[threadIdx.x];
    if (c)
        atomicAdd(&ctl->hist[pass][threadIdx.x], c);
}

__global__ __launch_bounds__(256) void lex_collect_kernel(const uint64_t *__restrict__ keys, LexControl *__restrict__ ctl,
                                                          uint32_t limit, uint64_t *__restrict__ sel)
{
    uint64_t kth;
    uint32_t k_rem;
    lex_derive_state(ctl, limit, kPasses, &kth, &k_rem); // all 64 bits fixed: the k-th largest key itself
    const uint32_t n = ctl->n_touched;
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t i0 = blockIdx.x * 256; i0 < n; i0 += stride) {
        const uint32_t i = i0 + threadIdx.x;
        const uint64_t key = i < n ? keys[i] : 0ull;
        const bool take = i < n && key >= kth;
        const uint32_t slot = wave_append_slot(take, &ctl->n_sel);
        if (take && slot < kMaxLimit)
            sel[slot] = key;
    }
}

// ---- sampled selection: 3 launches instead of 8 radix passes + collect ------------------------------------------
// The `limit` best of n_touched documents by (score desc, row asc), for limit << n_touched:
//   lex_sample_kernel   one workgroup reads a strided sample of 4096 or 8192 (score, row) keys and takes, by an LDS radix
//                       select, the sample's r-th largest key as threshold, r = mu + 4.5 sqrt(mu) + 8 with mu =
//                       limit * s / n the expected number of sample members among the true top `limit`: the threshold
//                       lies at or below the true limit-th key unless the sample holds > r of them (< 1e-5).  Full
//                       64-bit keys, not scores: chunks of equal length make BM25 scores tie by the thousand, and a
//                       threshold on the score alone would let all of a tie class through;
//   lex_filter_kernel   one pass over the touched documents: everything at or above the threshold is appended to a
//                       candidate list (~1.5 limit entries expected, capacity 8192);
//   lex_final_kernel    one workgroup: the exact limit-th largest 64-bit (score, row) key among the candidates (radix
//                       select in LDS) and everything at or above it -- exactly `limit` keys, unordered or sorted.
// When the list overflowed or came out short (the unlucky sample), the count
// word is set to kLexRetry and the caller repeats the query on the exact eight-pass path.
constexpr uint32_t kLexRetry = 0xFFFFFFFFu;
constexpr uint32_t kSampleMax = 8192, kFastLimitMax = 4096; // (the launch picks 4096 or 8192 sample keys: sample_log2)

__global__ __launch_bounds__(1024) void lex_sample_kernel(const float *__restrict__ scores, const uint32_t *__restrict__ touched,
                                                          LexControl *__restrict__ ctl, uint32_t limit, uint32_t r_forced,
                                                          uint32_t row_bits, uint32_t sample_log2)
{
    __shared__ uint64_t s_k[kSampleMax];
    __shared__ uint32_t s_hist[2048];
    __shared__ uint32_t s_pick[3];
    const uint32_t n = ctl->n_touched;
    const uint32_t s = min(n, 1u << sample_log2); // 4096 or 8192
    if (limit >= n || s == 0) { // everything is wanted
        if (threadIdx.x == 0)
            ctl->thr = 0;
        return;
    }
    // two dependent gathers per sample (list entry -> its score): all eight list loads of a thread first, then all
    // eight score loads, instead of eight serial round trips
    constexpr int kPer = kSampleMax / 1024;
    uint32_t rows_[kPer];
    float sc_[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const uint32_t i = threadIdx.x + 1024 * u;
        // strided: every part of the list (s is a power of two whenever it is not n itself: a shift, not a 64-bit division)
        const uint32_t at = i >= s ? 0u : s == n ? i : static_cast<uint32_t>((static_cast<uint64_t>(i) * n) >> sample_log2);
        rows_[u] = 0;
        if (1024u * u < s) // (uniform: the upper half of the slots is idle with the smaller sample)
            rows_[u] = touched[at];
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        sc_[u] = 0.0f;
        if (1024u * u < s)
            sc_[u] = scores[rows_[u]];
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const uint32_t i = threadIdx.x + 1024 * u;
        if (i < s)
            s_k[i] = pack_result(sc_[u], rows_[u]);
    }
    __syncthreads();
    const float mu = static_cast<float>(limit) * static_cast<float>(s) / static_cast<float>(n);
    const uint32_t r = r_forced ? min(s, r_forced) : min(s, static_cast<uint32_t>(mu + 4.5f * sqrtf(mu)) + 8u);
    const uint64_t thr = lds_kth_key64(s_k, s, r, s_hist, s_pick, 1024, /*slack=*/8, row_bits); // a few sample keys more: harmless
    if (threadIdx.x == 0)
        ctl->thr = thr;
}

__global__ __launch_bounds__(256) void lex_filter_kernel(const float *__restrict__ scores, const uint32_t *__restrict__ touched,
                                                         LexControl *__restrict__ ctl, uint64_t *__restrict__ cand)
{
    const uint32_t n = ctl->n_touched;
    const uint64_t thr = ctl->thr;
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t i0 = blockIdx.x * 256; i0 < n; i0 += stride) {
        const uint32_t i = i0 + threadIdx.x;
        uint64_t key = 0;
        bool take = false;
        if (i < n) {
            const uint32_t row = touched[i];
            key = pack_result(scores[row], row);
            take = key >= thr;
        }
        const uint32_t slot = wave_append_slot(take, &ctl->n_cand);
        if (take && slot < kMaxLimit)
            cand[slot] = key;
    }
}

template <bool SORTED>
__global__ __launch_bounds__(1024) void lex_final_kernel(const uint64_t *__restrict__ cand, const LexControl *__restrict__ ctl,
                                                         uint32_t limit, uint64_t *__restrict__ out_keys,
                                                         uint32_t *__restrict__ out_n, uint32_t row_bits)
{
    __shared__ uint64_t s[kMaxLimit];
    __shared__ uint64_t s_win[SORTED ? kFastLimitMax : 1]; // the winners, to be sorted (the unordered form writes them out directly)
    __shared__ uint32_t s_hist[2048];
    __shared__ uint32_t s_pick[3];
    __shared__ uint32_t s_out;
    const uint32_t m = ctl->n_cand, want = min(limit, ctl->n_touched);
    if (m > kMaxLimit || m < want) { // overflow or a short list (the sample misjudged)
        if (threadIdx.x == 0)
            *out_n = kLexRetry;
        return;
    }
    for (uint32_t i = threadIdx.x; i < m; i += 1024)
        s[i] = cand[i];
    if (threadIdx.x == 0)
        s_out = 0;
    __syncthreads();
    uint64_t kth = 0;
    if (m > want)
        kth = lds_kth_key64(s, m, want, s_hist, s_pick, 1024, 0, row_bits); // keys are unique: exactly `want` of them are >= kth
    for (uint32_t i = threadIdx.x; i < m; i += 1024) {
        const uint64_t v = s[i];
        if (v >= kth) {
            const uint32_t at = atomicAdd(&s_out, 1u);
            if constexpr (SORTED)
                s_win[at] = v;
            else
                out_keys[at] = v;
        }
    }
    if constexpr (SORTED) {
        uint32_t n_pad = 1;
        while (n_pad < want)
            n_pad <<= 1;
        for (uint32_t i = want + threadIdx.x; i < n_pad; i += 1024)
            s_win[i] = 0;
        __syncthreads();
        lds_sort_desc(s_win, n_pad);
        for (uint32_t i = threadIdx.x; i < want; i += 1024)
            out_keys[i] = s_win[i];
    }
    if (threadIdx.x == 0)
        *out_n = want;
}

// ... and zeroes the control block the NEXT call of this workspace will use (`next`: not the one this call worked in)
__global__ __launch_bounds__(256) void lex_clear_kernel(float *__restrict__ scores, const uint32_t *__restrict__ touched,
                                                        const LexControl *__restrict__ ctl, uint32_t *__restrict__ next,
                                                        uint32_t next_words)
{
    const uint32_t n = ctl->n_touched;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        scores[touched[i]] = 0.0f;
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < next_words; i += 256)
            next[i] = 0u;
}

// ---- removal of rows without rebuilding the postings on the host ---------------------------------------------
// rlr_lexical_remove_rows renumbers the surviving rows (the compaction rlr_index_delete_rows applies to the matrix).
// On the device that is one ordered stream compaction per posting segment: drop the postings of removed rows, map
// the others through remap[old row] -> new row, keep the order -- the CSR stays valid with the per-term counts the
// host already maintains.  Three launches: survivors per 2048-posting block, exclusive scan of the block counts,
// ordered scatter.
constexpr uint32_t kDeadRow = 0xFFFFFFFFu;
constexpr uint32_t kCsrBlock = 2048;

__global__ __launch_bounds__(256) void csr_count_kernel(const uint32_t *__restrict__ post_row, uint64_t total,
                                                        const uint32_t *__restrict__ remap, uint32_t *__restrict__ block_cnt)
{
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0)
        s_cnt = 0;
    __syncthreads();
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kCsrBlock;
    uint32_t mine = 0;
    for (uint32_t j = 0; j < kCsrBlock / 256; ++j) {
        const uint64_t i = base + j * 256 + threadIdx.x;
        if (i < total && remap[post_row[i]] != kDeadRow)
            ++mine;
    }
    if (mine)
        atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0)
        block_cnt[blockIdx.x] = s_cnt;
}

// in-place exclusive scan of nb counts by one workgroup (chunks of 1024 with a running carry)
__global__ __launch_bounds__(1024) void csr_scan_kernel(uint32_t *__restrict__ v, uint32_t nb)
{
    __shared__ uint32_t s[1024];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0)
        s_carry = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < nb; c0 += 1024) {
        const uint32_t i = c0 + threadIdx.x;
        const uint32_t x = i < nb ? v[i] : 0u;
        s[threadIdx.x] = x;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan
            const uint32_t add = threadIdx.x >= d ? s[threadIdx.x - d] : 0u;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        const uint32_t carry = s_carry;
        if (i < nb)
            v[i] = carry + s[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == 1023)
            s_carry = carry + s[1023];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void csr_scatter_kernel(const uint32_t *__restrict__ post_row, const uint32_t *__restrict__ post_tf,
                                                          uint64_t total, const uint32_t *__restrict__ remap,
                                                          const uint32_t *__restrict__ block_off, uint32_t *__restrict__ out_row,
                                                          uint32_t *__restrict__ out_tf)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base;
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kCsrBlock;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        s_base = block_off[blockIdx.x];
    __syncthreads();
    for (uint32_t j = 0; j < kCsrBlock / 256; ++j) { // sub-chunks in order, threads in order inside one: a stable compaction
        const uint64_t i = base + j * 256 + threadIdx.x;
        uint32_t nr = kDeadRow, tf = 0;
        if (i < total) {
            nr = remap[post_row[i]];
            tf = post_tf[i];
        }
        const bool keep = nr != kDeadRow;
        const uint64_t mask = __ballot(keep);
        if (lane == 0)
            s_wave[wave] = static_cast<uint32_t>(__popcll(mask));
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; ++w)
            before += s_wave[w];
        const uint32_t chunk_total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        const uint32_t at = s_base + before + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
        if (keep) {
            out_row[at] = nr;
            out_tf[at] = tf;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            s_base += chunk_total;
        __syncthreads();
    }
}

template <typename T>
int32_t dev_grow(T **p, uint64_t *cap, uint64_t need, bool zero = false)
{
    if (*cap >= need && *p)
        return RLR_OK;
    if (*p)
        (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const uint64_t n = std::max<uint64_t>(need, 1024);
    LEX_HIP(rlr::dev_malloc(reinterpret_cast<void **>(p), n * sizeof(T)));
    if (zero) { // null stream, then wait: the scoring stream is non-blocking and not ordered against it
        LEX_HIP(hipMemset(*p, 0, n * sizeof(T)));
        LEX_HIP(hipStreamSynchronize(nullptr));
    }
    *cap = n;
    return RLR_OK;
}

void split_tokens(const char *s, size_t len, std::vector<std::string> *out)
{
    size_t i = 0;
    while (i < len) {
        while (i < len && s[i] == ' ')
            ++i;
        size_t j = i;
        while (j < len && s[j] != ' ')
            ++j;
        if (j > i)
            out->emplace_back(s + i, j - i);
        i = j;
    }
}

} // namespace

// One scoring call's device state: its own stream, dense accumulator and selection buffers, so calls on different host
// threads run side by side (the reference's engine sits behind a tokio RwLock and serves reads concurrently).
struct LexWorkspace {
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr; // recorded behind a call's last kernel (consumers on other streams wait on it)
    float *d_scores = nullptr; // one f32 per row, all zero between calls
    uint64_t scores_cap = 0;
    uint32_t *d_touched = nullptr;
    uint64_t touched_cap = 0;
    uint64_t *d_keys = nullptr;
    uint64_t keys_cap = 0;
    uint64_t *d_sel = nullptr;
    LexControl *d_ctl = nullptr; // TWO control blocks: a call works in d_ctl[ctl_cur] and its clean-up launch (behind `ready`, off
                                 // the critical path) zeroes the other one for the next call -- it was a launch in front of every call
    uint32_t ctl_cur = 0;
    uint64_t *d_out = nullptr; // kMaxLimit keys + count
    uint64_t *h_out = nullptr; // pinned mirror
    bool dirty = false;        // a call failed after enqueuing work: accumulators / control may be non-zero
};

constexpr int kMaxWorkspaces = 8; // callers beyond this wait for a free one

struct rlr_lexical {
    int32_t device = 0;
    int n_cu = 256;
    bool terms_global = false; // RLR_LEX_TERMS=global at creation: the BM25 sums always in device memory (bm25_terms_kernel)
    uint32_t lds_wgs_forced = 0; // RLR_LEX_LDS_WGS=n at creation (a test switch): workgroups of bm25_terms_lds_kernel -- few of
                                 // them give each one more rows than a small test corpus would (several chunks per term)
    std::shared_mutex mu; // mutators and commit exclusive, scoring calls shared
    // ---- host state (LexicalIndex fields, rag_engine.rs:2084-2090, keyed by row instead of chunk id)
    std::unordered_map<std::string, uint32_t> term_id;
    std::vector<uint32_t> df;                                          // term -> documents holding it
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> doc_terms; // row -> (term, count); empty = absent
    std::vector<uint32_t> doc_len;                                     // row -> token count
    uint64_t total_docs = 0, total_length = 0, n_postings = 0, n_live_terms = 0;
    // Two posting segments on the device, both CSR by term.  MAIN holds rows [0, main_rows) as of the last full
    // rebuild; DELTA holds the rows appended since (row >= main_rows) and is the only thing rebuilt when a commit finds
    // nothing but appends -- an ingest loop that searches after every document would otherwise re-upload the whole
    // corpus' postings each time.  Every row lives in exactly one segment, so a term still adds to a row once per
    // launch and the f32 sum order is unchanged.  DELTA is folded into MAIN when it outgrows main / 8 (amortised O(1)).
    bool full_dirty = true;   // a row of MAIN changed (replace / remove / clear), or nothing was built yet
    bool delta_dirty = false; // only rows >= main_rows changed since the last commit
    uint64_t main_rows = 0, main_postings = 0, delta_postings = 0, n_full_commits = 0, n_delta_commits = 0;
    std::atomic<uint64_t> n_select_retries{0}; // queries the sampled selection handed back to the exact path
    std::vector<uint32_t> main_df; // documents per term inside MAIN (terms born later: beyond its end, 0)
    // ---- device CSR
    std::vector<uint64_t> term_off, dterm_off; // MAIN / DELTA offsets by term
    uint32_t *d_post_row = nullptr, *d_post_tf = nullptr, *d_doc_len = nullptr;
    uint32_t *d_dpost_row = nullptr, *d_dpost_tf = nullptr;
    uint64_t post_cap = 0, post_tf_cap = 0, doc_cap = 0, dpost_cap = 0, dpost_tf_cap = 0;
    uint32_t *d_remap = nullptr, *d_blocks = nullptr; // row removal on the device: old row -> new row, block counts
    uint64_t remap_cap = 0, blocks_cap = 0;
    uint64_t n_device_removals = 0;
    // ---- per-call workspaces
    std::mutex ws_mu;
    std::condition_variable ws_cv;
    std::vector<LexWorkspace *> ws_free;
    int ws_made = 0;
};

namespace {

void remove_row_stats(rlr_lexical *lx, uint64_t row)
{
    auto &terms = lx->doc_terms[row];
    if (terms.empty())
        return;
    const bool in_main = !lx->full_dirty && row < lx->main_rows; // (main_df is rebuilt anyway once full_dirty is set)
    for (const auto &tc : terms) {
        if (lx->df[tc.first] > 0 && --lx->df[tc.first] == 0)
            lx->n_live_terms--;
        if (in_main && tc.first < lx->main_df.size() && lx->main_df[tc.first] > 0)
            lx->main_df[tc.first]--;
    }
    lx->n_postings -= terms.size();
    const uint32_t len = lx->doc_len[row];
    lx->total_length = lx->total_length >= len ? lx->total_length - len : 0; // :2153-2157
    if (lx->total_docs > 0)
        lx->total_docs--;
    if (lx->total_docs == 0)
        lx->total_length = 0; // :2164-2166
    terms.clear();
    lx->doc_len[row] = 0;
}

void workspace_destroy(LexWorkspace *ws)
{
    if (!ws)
        return;
    if (ws->stream) {
        (void)hipStreamSynchronize(ws->stream);
        (void)hipStreamDestroy(ws->stream);
    }
    if (ws->ready)
        (void)hipEventDestroy(ws->ready);
    void *dev[] = {ws->d_scores, ws->d_touched, ws->d_keys, ws->d_sel, ws->d_ctl, ws->d_out};
    for (void *p : dev)
        if (p)
            (void)hipFree(p);
    if (ws->h_out)
        (void)hipHostFree(ws->h_out);
    delete ws;
}

int32_t workspace_create(LexWorkspace **out)
{
    LexWorkspace *ws = new (std::nothrow) LexWorkspace();
    if (!ws)
        return set_error(RLR_E_OOM, "host allocation failed");
    hipError_t e = hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&ws->ready, hipEventDisableTiming);
    if (e == hipSuccess)
        e = rlr::dev_malloc(reinterpret_cast<void **>(&ws->d_ctl), 2 * sizeof(LexControl));
    if (e == hipSuccess)
        e = hipMemsetAsync(ws->d_ctl, 0, 2 * sizeof(LexControl), ws->stream); // ordered before every call on this stream
    if (e == hipSuccess)
        e = rlr::dev_malloc(reinterpret_cast<void **>(&ws->d_sel), kMaxLimit * sizeof(uint64_t));
    if (e == hipSuccess)
        e = rlr::dev_malloc(reinterpret_cast<void **>(&ws->d_out), (kMaxLimit + 1) * sizeof(uint64_t));
    if (e == hipSuccess)
        e = hipHostMalloc(reinterpret_cast<void **>(&ws->h_out), (kMaxLimit + 1) * sizeof(uint64_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        workspace_destroy(ws);
        return set_error(e == hipErrorOutOfMemory ? RLR_E_OOM : RLR_E_HIP, "lexical workspace setup failed: %s",
                         hipGetErrorString(e));
    }
    *out = ws;
    return RLR_OK;
}

// A free workspace, a new one while fewer than kMaxWorkspaces exist, else wait for a call to finish.
int32_t workspace_acquire(rlr_lexical *lx, LexWorkspace **out)
{
    std::unique_lock<std::mutex> lk(lx->ws_mu);
    for (;;) {
        if (!lx->ws_free.empty()) {
            *out = lx->ws_free.back();
            lx->ws_free.pop_back();
            return RLR_OK;
        }
        if (lx->ws_made < kMaxWorkspaces) {
            lx->ws_made++;
            lk.unlock();
            const int32_t st = workspace_create(out);
            if (st != RLR_OK) {
                lk.lock();
                lx->ws_made--;
                lx->ws_cv.notify_one();
            }
            return st;
        }
        lx->ws_cv.wait(lk);
    }
}

int32_t commit_full(rlr_lexical *lx)
{
    const uint64_t n_rows = lx->doc_terms.size();
    const size_t n_terms = lx->df.size();
    lx->term_off.assign(n_terms + 1, 0);
    for (size_t t = 0; t < n_terms; ++t)
        lx->term_off[t + 1] = lx->term_off[t] + lx->df[t];
    const uint64_t total = lx->term_off[n_terms];
    std::vector<uint32_t> rows(std::max<uint64_t>(total, 1)), tfs(std::max<uint64_t>(total, 1));
    std::vector<uint64_t> fill(lx->term_off.begin(), lx->term_off.end() - 1);
    for (uint64_t r = 0; r < n_rows; ++r)
        for (const auto &tc : lx->doc_terms[r]) { // rows ascending -> every posting list is sorted by row
            const uint64_t at = fill[tc.first]++;
            rows[at] = static_cast<uint32_t>(r);
            tfs[at] = tc.second;
        }
    LEX_TRY(dev_grow(&lx->d_post_row, &lx->post_cap, total));
    LEX_TRY(dev_grow(&lx->d_post_tf, &lx->post_tf_cap, total));
    LEX_TRY(dev_grow(&lx->d_doc_len, &lx->doc_cap, n_rows + n_rows / 4)); // headroom: appends upload only their part
    if (total) {
        LEX_HIP(hipMemcpy(lx->d_post_row, rows.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice));
        LEX_HIP(hipMemcpy(lx->d_post_tf, tfs.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (n_rows)
        LEX_HIP(hipMemcpy(lx->d_doc_len, lx->doc_len.data(), n_rows * sizeof(uint32_t), hipMemcpyHostToDevice));
    lx->main_rows = n_rows;
    lx->main_postings = total;
    lx->main_df = lx->df;
    lx->dterm_off.clear();
    lx->delta_postings = 0;
    lx->full_dirty = lx->delta_dirty = false;
    lx->n_full_commits++;
    return RLR_OK;
}

// Only rows >= main_rows changed: rebuild DELTA from them (O(terms + their postings)), upload their lengths.
int32_t commit_delta(rlr_lexical *lx)
{
    const uint64_t n_rows = lx->doc_terms.size();
    const size_t n_terms = lx->df.size();
    lx->dterm_off.assign(n_terms + 1, 0);
    for (size_t t = 0; t < n_terms; ++t) {
        const uint32_t in_main = t < lx->main_df.size() ? lx->main_df[t] : 0u;
        lx->dterm_off[t + 1] = lx->dterm_off[t] + (lx->df[t] - in_main); // MAIN is untouched, so the rest is DELTA's
    }
    const uint64_t total = lx->dterm_off[n_terms];
    std::vector<uint32_t> rows(std::max<uint64_t>(total, 1)), tfs(std::max<uint64_t>(total, 1));
    std::vector<uint64_t> fill(lx->dterm_off.begin(), lx->dterm_off.end() - 1);
    for (uint64_t r = lx->main_rows; r < n_rows; ++r)
        for (const auto &tc : lx->doc_terms[r]) {
            const uint64_t at = fill[tc.first]++;
            rows[at] = static_cast<uint32_t>(r);
            tfs[at] = tc.second;
        }
    LEX_TRY(dev_grow(&lx->d_dpost_row, &lx->dpost_cap, total + total / 2));
    LEX_TRY(dev_grow(&lx->d_dpost_tf, &lx->dpost_tf_cap, total + total / 2));
    if (total) {
        LEX_HIP(hipMemcpy(lx->d_dpost_row, rows.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice));
        LEX_HIP(hipMemcpy(lx->d_dpost_tf, tfs.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (n_rows > lx->doc_cap || !lx->d_doc_len) { // outgrew the headroom: new buffer, every length again
        LEX_TRY(dev_grow(&lx->d_doc_len, &lx->doc_cap, n_rows + n_rows / 4));
        LEX_HIP(hipMemcpy(lx->d_doc_len, lx->doc_len.data(), n_rows * sizeof(uint32_t), hipMemcpyHostToDevice));
    } else if (n_rows > lx->main_rows) {
        LEX_HIP(hipMemcpy(lx->d_doc_len + lx->main_rows, lx->doc_len.data() + lx->main_rows,
                          (n_rows - lx->main_rows) * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    lx->delta_postings = total;
    lx->delta_dirty = false;
    lx->n_delta_commits++;
    return RLR_OK;
}

// caller holds the index exclusively
int32_t commit(rlr_lexical *lx)
{
    if (!lx->full_dirty) {
        const uint64_t pending = lx->n_postings - lx->main_postings; // what DELTA would hold
        if (pending <= std::max<uint64_t>(65536, lx->main_postings / 8))
            return commit_delta(lx);
    }
    return commit_full(lx);
}

// Bring both posting segments, the per-term offsets and the document lengths in line with a removal the host state has
// already absorbed.  remap: old row -> new row (kDeadRow: removed).  Caller holds the index exclusively.
int32_t compact_segment(rlr_lexical *lx, uint32_t **rows, uint32_t **tfs, uint64_t *cap_rows, uint64_t *cap_tfs, uint64_t total,
                        uint64_t new_total)
{
    if (total == 0)
        return RLR_OK;
    const uint32_t nb = static_cast<uint32_t>((total + kCsrBlock - 1) / kCsrBlock);
    LEX_TRY(dev_grow(&lx->d_blocks, &lx->blocks_cap, nb));
    uint32_t *out_rows = nullptr, *out_tfs = nullptr;
    const uint64_t cap = std::max<uint64_t>(new_total + new_total / 8, 1024);
    LEX_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&out_rows), cap * sizeof(uint32_t)));
    hipError_t e = rlr::dev_malloc(reinterpret_cast<void **>(&out_tfs), cap * sizeof(uint32_t));
    if (e != hipSuccess) {
        (void)hipFree(out_rows);
        return set_error(RLR_E_OOM, "lexical compaction: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(csr_count_kernel, dim3(nb), dim3(256), 0, nullptr, *rows, total, lx->d_remap, lx->d_blocks);
    hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, nullptr, lx->d_blocks, nb);
    hipLaunchKernelGGL(csr_scatter_kernel, dim3(nb), dim3(256), 0, nullptr, *rows, *tfs, total, lx->d_remap, lx->d_blocks, out_rows,
                       out_tfs);
    e = hipGetLastError();
    if (e == hipSuccess)
        e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        (void)hipFree(out_rows);
        (void)hipFree(out_tfs);
        return set_error(RLR_E_HIP, "lexical compaction failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(*rows);
    (void)hipFree(*tfs);
    *rows = out_rows;
    *tfs = out_tfs;
    *cap_rows = *cap_tfs = cap;
    return RLR_OK;
}

int32_t remove_on_device(rlr_lexical *lx, const std::vector<uint32_t> &remap, uint64_t dead_main_rows, uint64_t dead_main_postings,
                         uint64_t dead_delta_postings)
{
    LEX_HIP(hipSetDevice(lx->device));
    LEX_TRY(dev_grow(&lx->d_remap, &lx->remap_cap, remap.size()));
    LEX_HIP(hipMemcpy(lx->d_remap, remap.data(), remap.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    const uint64_t new_main = lx->main_postings - dead_main_postings, new_delta = lx->delta_postings - dead_delta_postings;
    LEX_TRY(compact_segment(lx, &lx->d_post_row, &lx->d_post_tf, &lx->post_cap, &lx->post_tf_cap, lx->main_postings, new_main));
    LEX_TRY(compact_segment(lx, &lx->d_dpost_row, &lx->d_dpost_tf, &lx->dpost_cap, &lx->dpost_tf_cap, lx->delta_postings,
                            new_delta));
    // per-term offsets from the counts the host keeps (the compaction preserved the order inside every posting list)
    const size_t n_terms = lx->df.size();
    lx->main_df.resize(n_terms, 0);
    lx->term_off.assign(n_terms + 1, 0);
    lx->dterm_off.assign(n_terms + 1, 0);
    for (size_t t = 0; t < n_terms; ++t) {
        lx->term_off[t + 1] = lx->term_off[t] + lx->main_df[t];
        lx->dterm_off[t + 1] = lx->dterm_off[t] + (lx->df[t] - lx->main_df[t]);
    }
    if (lx->term_off[n_terms] != new_main || lx->dterm_off[n_terms] != new_delta)
        return set_error(RLR_E_HIP, "lexical compaction: posting counts disagree (%llu / %llu main, %llu / %llu appended)",
                         static_cast<unsigned long long>(lx->term_off[n_terms]), static_cast<unsigned long long>(new_main),
                         static_cast<unsigned long long>(lx->dterm_off[n_terms]), static_cast<unsigned long long>(new_delta));
    lx->main_rows -= dead_main_rows;
    lx->main_postings = new_main;
    lx->delta_postings = new_delta;
    const uint64_t n_rows = lx->doc_terms.size();
    if (n_rows > lx->doc_cap || !lx->d_doc_len)
        LEX_TRY(dev_grow(&lx->d_doc_len, &lx->doc_cap, n_rows + n_rows / 4));
    if (n_rows)
        LEX_HIP(hipMemcpy(lx->d_doc_len, lx->doc_len.data(), n_rows * sizeof(uint32_t), hipMemcpyHostToDevice));
    lx->n_device_removals++;
    return RLR_OK;
}

} // namespace

extern "C" {

int32_t rlr_lexical_create(int32_t device_id, rlr_lexical **out)
{
    if (!out)
        return set_error(RLR_E_INVALID, "out is null");
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return set_error(RLR_E_NO_DEVICE, "no HIP device: the lexical index has no CPU path");
    if (device_id < 0 || device_id >= n_dev)
        return set_error(RLR_E_NO_DEVICE, "device %d out of range (%d devices)", device_id, n_dev);
    rlr_lexical *lx = new (std::nothrow) rlr_lexical();
    if (!lx)
        return set_error(RLR_E_OOM, "host allocation failed");
    lx->device = device_id;
    hipDeviceProp_t prop;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess)
        e = hipGetDeviceProperties(&prop, device_id);
    if (e == hipSuccess) {
        lx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (const char *v = getenv("RLR_LEX_TERMS"))
            lx->terms_global = !strcmp(v, "global");
        if (const char *v = getenv("RLR_LEX_LDS_WGS"))
            lx->lds_wgs_forced = static_cast<uint32_t>(strtoul(v, nullptr, 0));
    }
    if (e != hipSuccess) {
        rlr_lexical_destroy(lx);
        return set_error(e == hipErrorOutOfMemory ? RLR_E_OOM : RLR_E_HIP, "lexical index setup failed: %s",
                         hipGetErrorString(e));
    }
    LexWorkspace *ws = nullptr; // the first workspace now, so a device without memory fails here and not in score
    const int32_t st = workspace_create(&ws);
    if (st != RLR_OK) {
        rlr_lexical_destroy(lx);
        return st;
    }
    lx->ws_free.push_back(ws);
    lx->ws_made = 1;
    *out = lx;
    return RLR_OK;
}

void rlr_lexical_destroy(rlr_lexical *lx)
{
    if (!lx)
        return;
    (void)hipSetDevice(lx->device);
    for (LexWorkspace *ws : lx->ws_free) // every call has returned (the caller's contract), so all of them are here
        workspace_destroy(ws);
    void *dev[] = {lx->d_post_row, lx->d_post_tf, lx->d_doc_len, lx->d_dpost_row, lx->d_dpost_tf, lx->d_remap, lx->d_blocks};
    for (void *p : dev)
        if (p)
            (void)hipFree(p);
    delete lx;
}

int32_t rlr_lexical_add_chunk(rlr_lexical *lx, uint64_t row, const char *tokens, size_t len)
{
    if (!lx)
        return set_error(RLR_E_INVALID, "lexical handle is null");
    if (len && !tokens)
        return set_error(RLR_E_INVALID, "tokens is null");
    if (row >= 0xFFFFFFFFull)
        return set_error(RLR_E_RANGE, "row %llu does not fit the 32-bit postings", static_cast<unsigned long long>(row));
    std::unique_lock<std::shared_mutex> lk(lx->mu);
    if (row >= lx->doc_terms.size()) {
        lx->doc_terms.resize(row + 1);
        lx->doc_len.resize(row + 1, 0);
    }
    remove_row_stats(lx, row); // `if self.doc_terms.contains_key(id) { self.remove_chunk(id) }`
    if (row < lx->main_rows)
        lx->full_dirty = true; // a row of the main segment changes
    else
        lx->delta_dirty = true;
    std::vector<std::string> toks;
    split_tokens(tokens, len, &toks);
    if (toks.empty())
        return RLR_OK; // :2112-2114
    std::unordered_map<uint32_t, uint32_t> counts;
    std::vector<uint32_t> order; // first-occurrence order keeps the per-row list reproducible
    for (const auto &t : toks) {
        auto it = lx->term_id.find(t);
        uint32_t id;
        if (it == lx->term_id.end()) {
            id = static_cast<uint32_t>(lx->df.size());
            lx->term_id.emplace(t, id);
            lx->df.push_back(0);
        } else {
            id = it->second;
        }
        if (counts[id]++ == 0)
            order.push_back(id);
    }
    auto &terms = lx->doc_terms[row];
    terms.reserve(order.size());
    uint32_t doc_length = 0;
    for (uint32_t id : order) {
        terms.emplace_back(id, counts[id]);
        doc_length += counts[id];
        if (lx->df[id]++ == 0)
            lx->n_live_terms++;
    }
    lx->n_postings += terms.size();
    lx->doc_len[row] = doc_length;
    lx->total_docs += 1;
    lx->total_length += doc_length;
    return RLR_OK;
}

int32_t rlr_lexical_remove_rows(rlr_lexical *lx, const uint64_t *rows, uint32_t n)
{
    if (!lx)
        return set_error(RLR_E_INVALID, "lexical handle is null");
    if (n == 0)
        return RLR_OK;
    if (!rows)
        return set_error(RLR_E_INVALID, "rows is null");
    std::unique_lock<std::shared_mutex> lk(lx->mu);
    const uint64_t size = lx->doc_terms.size();
    // The device postings can follow the removal by a compaction (no host rebuild, no re-upload) when they are current;
    // pending appends are committed first (cheap: the appended segment only).
    bool on_device = !lx->full_dirty && (lx->main_postings > 0 || lx->delta_postings > 0 || lx->delta_dirty);
    if (on_device && lx->delta_dirty) {
        LEX_HIP(hipSetDevice(lx->device));
        LEX_TRY(commit_delta(lx));
    }
    std::vector<char> dead(size, 0);
    bool any = false;
    uint64_t dead_main_rows = 0, dead_main_postings = 0, dead_delta_postings = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (rows[i] < size && !dead[rows[i]]) {
            dead[rows[i]] = 1;
            if (on_device) {
                if (rows[i] < lx->main_rows) {
                    dead_main_rows++;
                    dead_main_postings += lx->doc_terms[rows[i]].size();
                } else {
                    dead_delta_postings += lx->doc_terms[rows[i]].size();
                }
            }
            remove_row_stats(lx, rows[i]); // (also takes the row's terms out of main_df)
            any = true;
        }
    if (!any)
        return RLR_OK;
    std::vector<uint32_t> remap;
    if (on_device)
        remap.assign(size, kDeadRow);
    uint64_t w = 0; // stable compaction, the same renumbering rlr_index_delete_rows applies
    for (uint64_t r = 0; r < size; ++r)
        if (!dead[r]) {
            if (w != r) {
                lx->doc_terms[w] = std::move(lx->doc_terms[r]);
                lx->doc_len[w] = lx->doc_len[r];
            }
            if (on_device)
                remap[r] = static_cast<uint32_t>(w);
            ++w;
        }
    lx->doc_terms.resize(w);
    lx->doc_len.resize(w);
    if (!on_device || lx->main_postings >= 0xFFFFFFFFull || lx->delta_postings >= 0xFFFFFFFFull) {
        lx->full_dirty = true; // rows are renumbered: the next score call rebuilds the postings
        return RLR_OK;
    }
    const int32_t st = remove_on_device(lx, remap, dead_main_rows, dead_main_postings, dead_delta_postings);
    if (st != RLR_OK)
        lx->full_dirty = true; // whatever state the device arrays are in, the next commit replaces them
    return st;
}

int32_t rlr_lexical_clear(rlr_lexical *lx)
{
    if (!lx)
        return set_error(RLR_E_INVALID, "lexical handle is null");
    std::unique_lock<std::shared_mutex> lk(lx->mu);
    lx->term_id.clear();
    lx->df.clear();
    lx->doc_terms.clear();
    lx->doc_len.clear();
    lx->total_docs = lx->total_length = lx->n_postings = lx->n_live_terms = 0;
    lx->full_dirty = true;
    return RLR_OK;
}

int32_t rlr_lexical_contains(rlr_lexical *lx, uint64_t row)
{
    if (!lx)
        return set_error(RLR_E_INVALID, "lexical handle is null");
    std::shared_lock<std::shared_mutex> lk(lx->mu);
    return row < lx->doc_terms.size() && !lx->doc_terms[row].empty() ? 1 : 0;
}

int32_t rlr_lexical_info(rlr_lexical *lx, uint64_t *total_docs, uint64_t *total_length, uint64_t *n_terms,
                         uint64_t *n_postings)
{
    if (!lx)
        return set_error(RLR_E_INVALID, "lexical handle is null");
    std::shared_lock<std::shared_mutex> lk(lx->mu);
    if (total_docs) *total_docs = lx->total_docs;
    if (total_length) *total_length = lx->total_length;
    if (n_terms) *n_terms = lx->n_live_terms;
    if (n_postings) *n_postings = lx->n_postings;
    return RLR_OK;
}

int32_t rlr_lexical_segments(rlr_lexical *lx, uint64_t *main_postings, uint64_t *appended_postings, uint64_t *full_rebuilds,
                             uint64_t *append_rebuilds, uint64_t *select_retries)
{
    if (!lx)
        return set_error(RLR_E_INVALID, "lexical handle is null");
    std::shared_lock<std::shared_mutex> lk(lx->mu);
    if (main_postings) *main_postings = lx->main_postings;
    if (appended_postings) *appended_postings = lx->delta_postings;
    if (full_rebuilds) *full_rebuilds = lx->n_full_commits;
    if (append_rebuilds) *append_rebuilds = lx->n_delta_commits;
    if (select_retries) *select_retries = lx->n_select_retries.load();
    return RLR_OK;
}

// first_attempt = 1: skip the sampled selection (a fused search already saw it hand this query back)
static int32_t lexical_score_from(rlr_lexical *lx, const char *query_tokens, size_t len, uint32_t limit, uint64_t *rows_out,
                                  float *scores_out, uint32_t *n_out, int first_attempt);

int32_t rlr_lexical_score(rlr_lexical *lx, const char *query_tokens, size_t len, uint32_t limit, uint64_t *rows_out,
                          float *scores_out, uint32_t *n_out)
{
    return lexical_score_from(lx, query_tokens, len, limit, rows_out, scores_out, n_out, 0);
}

extern "C++" {
namespace rlr {
int32_t lexical_score_exact(rlr_lexical *lx, const char *query_tokens, size_t len, uint32_t limit, uint64_t *rows_out,
                            float *scores_out, uint32_t *n_out)
{
    if (lx)
        lx->n_select_retries++; // the fused search's sampled selection gave up on this query
    return lexical_score_from(lx, query_tokens, len, limit, rows_out, scores_out, n_out, 1);
}
} // namespace rlr
} // extern "C++"

static int32_t lexical_score_from(rlr_lexical *lx, const char *query_tokens, size_t len, uint32_t limit, uint64_t *rows_out,
                                  float *scores_out, uint32_t *n_out, int first_attempt)
{
    if (!lx)
        return set_error(RLR_E_INVALID, "lexical handle is null");
    if (!n_out)
        return set_error(RLR_E_INVALID, "n_out is null");
    *n_out = 0;
    if (len && !query_tokens)
        return set_error(RLR_E_INVALID, "query_tokens is null");
    for (int attempt = first_attempt; attempt < 2; ++attempt) {
        rlr::LexPending p;
        LEX_TRY(rlr::lexical_enqueue(lx, query_tokens, len, limit, &p, /*need_sorted=*/true, /*exact_passes=*/attempt == 1));
        if (p.limit == 0) // empty index, no tokens, or no term of the query is known (:2170-2177, :2196)
            return RLR_OK;
        int32_t st = RLR_OK;
        if (!rows_out || !scores_out)
            st = set_error(RLR_E_INVALID, "rows_out / scores_out is null");
        else
            st = rlr::lexical_fetch(&p, rows_out, scores_out, n_out);
        rlr::lexical_finish(&p, st == RLR_OK);
        if (st != RLR_OK || *n_out != kLexRetry)
            return st;
        *n_out = 0; // the sampled selection handed the query back: once more with the exact radix passes
        lx->n_select_retries++;
    }
    return set_error(RLR_E_HIP, "lexical selection did not converge");
}

int32_t rlr_tokenize_ascii(const char *text, size_t len, char *out, size_t cap, size_t *out_len)
{
    if ((len && !text) || !out_len)
        return set_error(RLR_E_INVALID, "null argument");
    size_t w = 0;
    bool first = true;
    size_t i = 0;
    auto alnum = [](unsigned char c) { return c >= 0x80 || (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); };
    while (i < len) {
        while (i < len && !alnum(static_cast<unsigned char>(text[i])))
            ++i;
        size_t j = i;
        while (j < len && alnum(static_cast<unsigned char>(text[j])))
            ++j;
        if (j - i >= 3) { // `token.len() >= 3`: bytes
            if (!first) {
                if (out && w < cap)
                    out[w] = ' ';
                ++w;
            }
            for (size_t p = i; p < j; ++p) {
                const unsigned char c = static_cast<unsigned char>(text[p]);
                if (out && w < cap)
                    out[w] = static_cast<char>((c >= 'A' && c <= 'Z') ? c + 32 : c);
                ++w;
            }
            first = false;
        }
        i = j;
    }
    *out_len = w;
    if (w > cap)
        return set_error(RLR_E_RANGE, "tokenize: output needs %zu bytes, capacity is %zu", w, cap);
    return RLR_OK;
}

} // extern "C"

namespace rlr {

namespace {
struct PendingGuard { // releases whatever lexical_enqueue had taken when it fails half way
    LexPending *p;
    bool armed = true;
    ~PendingGuard()
    {
        if (armed)
            lexical_finish(p, false);
    }
};
} // namespace

// expected candidates of the sampled selection: (mu + 4.5 sqrt(mu) + 8) n / s with mu = limit s / n the expected number of
// sample members among the true top `limit`
static double sampled_candidates(uint32_t limit, uint64_t n, uint32_t s)
{
    return static_cast<double>(limit) + 4.5 * std::sqrt(static_cast<double>(limit) * n / s) + 8.0 * static_cast<double>(n) / s;
}

int32_t lexical_enqueue(rlr_lexical *lx, const char *query_tokens, size_t len, uint32_t limit, LexPending *out,
                        bool need_sorted, bool exact_passes, const LexSink *sink)
{
    *out = LexPending{};
    out->lx = lx;
    std::vector<std::string> toks;
    split_tokens(query_tokens, len, &toks);
    if (toks.empty()) // :2175-2177
        return RLR_OK;
    const uint32_t lim = limit == 0 ? kMaxLimit : std::min(limit, kMaxLimit);
    // Scoring calls share the index; only rebuilding the device postings after a mutation needs it alone.
    lx->mu.lock_shared();
    out->locked = true;
    PendingGuard guard{out};
    if (lx->total_docs == 0) { // :2170-2172
        lexical_finish(out, true);
        guard.armed = false;
        return RLR_OK;
    }
    LEX_HIP(hipSetDevice(lx->device));
    while (lx->full_dirty || lx->delta_dirty) {
        lx->mu.unlock_shared();
        out->locked = false;
        {
            std::unique_lock<std::shared_mutex> wr(lx->mu);
            if (lx->full_dirty || lx->delta_dirty)
                LEX_TRY(commit(lx));
        }
        lx->mu.lock_shared();
        out->locked = true;
    }
    // unique query terms in order of first occurrence (:2179-2182 uses a HashSet: order unspecified there)
    std::vector<uint32_t> terms;
    for (const auto &t : toks) {
        auto it = lx->term_id.find(t);
        if (it == lx->term_id.end() || lx->df[it->second] == 0)
            continue; // `if let Some(postings) = self.term_postings.get(&term)` :2196
        if (std::find(terms.begin(), terms.end(), it->second) == terms.end())
            terms.push_back(it->second);
    }
    uint64_t upper = 0;
    for (uint32_t t : terms)
        upper += lx->df[t];
    if (upper == 0 || lx->total_docs == 0) { // (or emptied by another thread while the lock was released)
        lexical_finish(out, true);
        guard.armed = false;
        return RLR_OK;
    }
    upper = std::min<uint64_t>(upper, lx->doc_terms.size()); // at most one touched entry per row
    LexWorkspace *ws = nullptr;
    LEX_TRY(workspace_acquire(lx, &ws));
    out->ws = ws;
    const uint64_t n_rows = lx->doc_terms.size();
    if (ws->scores_cap < n_rows || !ws->d_scores) { // the index grew since this workspace last ran
        LEX_HIP(hipStreamSynchronize(ws->stream));
        LEX_TRY(dev_grow(&ws->d_scores, &ws->scores_cap, n_rows + n_rows / 4, /*zero=*/true));
        ws->dirty = false;
        LEX_HIP(hipMemsetAsync(ws->d_ctl, 0, 2 * sizeof(LexControl), ws->stream));
    }
    if (ws->dirty) { // restore the all-zero invariant a failed call may have broken
        LEX_HIP(hipStreamSynchronize(ws->stream));
        LEX_HIP(hipMemsetAsync(ws->d_scores, 0, ws->scores_cap * sizeof(float), ws->stream));
        LEX_HIP(hipMemsetAsync(ws->d_ctl, 0, 2 * sizeof(LexControl), ws->stream));
        ws->dirty = false;
    }
    LEX_TRY(dev_grow(&ws->d_touched, &ws->touched_cap, upper));

    const float n_docs = static_cast<float>(lx->total_docs);
    const float avg = static_cast<float>(lx->total_length) / n_docs; // :2184-2188
    hipStream_t s = ws->stream;
    ws->dirty = true; // cleared by lexical_finish(ok) once the whole pipeline has run
    // The control block (counters, histograms) of THIS call: all zero on entry -- the previous call's clean-up launch (or the
    // memsets above) saw to that -- and readable for a consumer on another stream until the workspace is handed back: the
    // next call works in the other block and zeroes this one only behind its own `ready`.
    static_assert(sizeof(LexControl) % 4 == 0, "cleared as 32-bit words");
    LexControl *ctl = ws->d_ctl + ws->ctl_cur;
    LexControl *ctl_next = ws->d_ctl + (ws->ctl_cur ^ 1u);
    ws->ctl_cur ^= 1u;
    const uint32_t max_blocks = static_cast<uint32_t>(lx->n_cu) * 8;
    static const bool per_term = getenv("RLR_LEX_PER_TERM") != nullptr; // (A/B switch: one launch per term and segment)
    // workgroups of the row-partitioned kernel: one per CU while each still owns a few hundred rows
    const uint32_t row_wgs = std::max<uint32_t>(1u, std::min<uint32_t>(static_cast<uint32_t>(lx->n_cu), static_cast<uint32_t>(n_rows / 256)));
    TermBatch tb{};
    // ... with the accumulators in LDS while a workgroup's rows fit there: one workgroup per 256 rows, up to 8 per CU
    // (RLR_LEX_TERMS=global when the index is created: the form that adds in device memory, also the one for larger indexes)
    const uint32_t lds_wgs = lx->lds_wgs_forced
                                 ? lx->lds_wgs_forced
                                 : static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(max_blocks, (n_rows + 255) / 256)));
    const bool terms_lds = !lx->terms_global && (n_rows + lds_wgs - 1) / lds_wgs + 1 <= kLdsRows;
    auto flush_terms = [&]() {
        if (tb.n_terms && terms_lds)
            hipLaunchKernelGGL(bm25_terms_lds_kernel, dim3(lds_wgs), dim3(256), 0, s, tb, lx->d_post_row, lx->d_post_tf,
                               lx->d_dpost_row, lx->d_dpost_tf, lx->d_doc_len, static_cast<uint32_t>(n_rows), avg, ws->d_scores,
                               ws->d_touched, ctl);
        else if (tb.n_terms)
            hipLaunchKernelGGL(bm25_terms_kernel, dim3(row_wgs), dim3(256), 0, s, tb, lx->d_post_row, lx->d_post_tf, lx->d_dpost_row,
                               lx->d_dpost_tf, lx->d_doc_len, static_cast<uint32_t>(n_rows), avg, ws->d_scores, ws->d_touched,
                               ctl);
        tb.n_terms = 0;
    };
    for (uint32_t t : terms) {
        const float df = static_cast<float>(lx->df[t]);
        float idf = std::log((n_docs - df + 0.5f) / (df + 0.5f)); // f32 ln (:2198-2200)
        idf = idf > 0.0f ? idf : 0.0f;                            // f32::max(0.0): NaN -> 0
        // the term's postings in the main segment, then in the appended one (a row is in exactly one of them)
        const uint32_t cnt_m = t < lx->main_df.size() ? lx->main_df[t] : 0u;
        const uint32_t cnt_d = lx->dterm_off.empty() ? 0u : static_cast<uint32_t>(lx->dterm_off[t + 1] - lx->dterm_off[t]);
        if (!per_term) {
            const uint32_t at = tb.n_terms++;
            tb.cnt_m[at] = cnt_m;
            tb.off_m[at] = cnt_m ? lx->term_off[t] : 0;
            tb.cnt_d[at] = cnt_d;
            tb.off_d[at] = cnt_d ? lx->dterm_off[t] : 0;
            tb.idf[at] = idf;
            if (tb.n_terms == kTermsPerLaunch)
                flush_terms(); // (more than 16 terms: the next launch continues in term order)
            continue;
        }
        if (cnt_m) {
            const uint64_t off = lx->term_off[t];
            const uint32_t blocks = std::min<uint32_t>((cnt_m + 255) / 256, max_blocks);
            hipLaunchKernelGGL(bm25_term_kernel, dim3(blocks), dim3(256), 0, s, lx->d_post_row + off, lx->d_post_tf + off, cnt_m,
                               lx->d_doc_len, avg, idf, ws->d_scores, ws->d_touched, ctl);
        }
        if (cnt_d) {
            const uint64_t off = lx->dterm_off[t];
            const uint32_t blocks = std::min<uint32_t>((cnt_d + 255) / 256, max_blocks);
            hipLaunchKernelGGL(bm25_term_kernel, dim3(blocks), dim3(256), 0, s, lx->d_dpost_row + off, lx->d_dpost_tf + off,
                               cnt_d, lx->d_doc_len, avg, idf, ws->d_scores, ws->d_touched, ctl);
        }
    }
    flush_terms();
    LEX_HIP(hipGetLastError());
    uint32_t *d_out_n = reinterpret_cast<uint32_t *>(ws->d_out + kMaxLimit);
    const uint32_t blocks_u = std::min<uint32_t>(static_cast<uint32_t>((upper + 255) / 256), max_blocks);
    const uint64_t *d_result = ws->d_out;
    const uint32_t *d_result_n = d_out_n;
    if (upper <= kMaxLimit) {
        hipLaunchKernelGGL(lex_sort_kernel<true>, dim3(1), dim3(1024), 0, s, ws->d_scores, ws->d_touched, nullptr, ctl,
                           lim, ws->d_out, d_out_n);
    } else if (!exact_passes && lim <= kFastLimitMax && sampled_candidates(lim, upper, kSampleMax) <= 6000.0) {
        // (beyond ~3 M touched documents even the 8192-entry sample is too coarse for the 8192-entry candidate list)
        // 4096 sample keys while that keeps the list short: half the gathers and LDS work of the sample launch for ~15 % more
        // candidates at 100 k touched documents
        const uint32_t sample_log2 = sampled_candidates(lim, upper, 4096) <= 4000.0 ? 12u : 13u;
        // sampled threshold -> one filter pass -> exact finish among the ~1.5 lim candidates (3 launches); the count word
        // says kLexRetry when that list overflowed or came out short
        // RLR_LEX_SAMPLE_RANK (a test switch): the sample rank to use instead of mu + 4.5 sqrt(mu) + 8 -- 1 makes the
        // threshold the largest sample key, the candidate list short, and every such query take the retry
        uint32_t row_bits = 0; // rows < 2^row_bits: the digits of the LDS radix selects skip the constant bits above
        while (row_bits < 32 && (n_rows - 1) >> row_bits)
            ++row_bits;
        static const uint32_t r_forced = getenv("RLR_LEX_SAMPLE_RANK") ? static_cast<uint32_t>(atoi(getenv("RLR_LEX_SAMPLE_RANK"))) : 0u;
        hipLaunchKernelGGL(lex_sample_kernel, dim3(1), dim3(1024), 0, s, ws->d_scores, ws->d_touched, ctl, lim, r_forced, row_bits,
                           sample_log2);
        hipLaunchKernelGGL(lex_filter_kernel, dim3(blocks_u), dim3(256), 0, s, ws->d_scores, ws->d_touched, ctl, ws->d_sel);
        if (need_sorted)
            hipLaunchKernelGGL(lex_final_kernel<true>, dim3(1), dim3(1024), 0, s, ws->d_sel, ctl, lim, ws->d_out, d_out_n, row_bits);
        else
            hipLaunchKernelGGL(lex_final_kernel<false>, dim3(1), dim3(1024), 0, s, ws->d_sel, ctl, lim, ws->d_out, d_out_n, row_bits);
        out->may_retry = true;
    } else {
        LEX_TRY(dev_grow(&ws->d_keys, &ws->keys_cap, upper));
        hipLaunchKernelGGL(lex_select_pass_kernel<true>, dim3(blocks_u), dim3(256), 0, s, ws->d_keys, ctl, lim, 0,
                           ws->d_scores, ws->d_touched);
        for (int p = 1; p < kPasses; ++p)
            hipLaunchKernelGGL(lex_select_pass_kernel<false>, dim3(blocks_u), dim3(256), 0, s, ws->d_keys, ctl, lim, p,
                               nullptr, nullptr);
        hipLaunchKernelGGL(lex_collect_kernel, dim3(blocks_u), dim3(256), 0, s, ws->d_keys, ctl, lim, ws->d_sel);
        if (need_sorted) {
            hipLaunchKernelGGL(lex_sort_kernel<false>, dim3(1), dim3(1024), 0, s, nullptr, nullptr, ws->d_sel, ctl, lim,
                               ws->d_out, d_out_n);
        } else { // the consumer (the hybrid blend) wants the set, not its order: one LDS sort less on the critical path
            d_result = ws->d_sel;
            d_result_n = &ctl->n_sel;
        }
    }
    if (sink && sink->d_rows) { // a hybrid search takes the result apart on its own stream otherwise: one launch behind its join
        launch_lex_unpack(d_result, d_result_n, std::min(lim, sink->n_bound), *sink, s);
        out->unpacked = true;
    }
    LEX_HIP(hipGetLastError());
    LEX_HIP(hipEventRecord(ws->ready, s)); // the result list is complete here; the clean-up below runs behind it
    hipLaunchKernelGGL(lex_clear_kernel, dim3(blocks_u), dim3(256), 0, s, ws->d_scores, ws->d_touched, ctl,
                       reinterpret_cast<uint32_t *>(ctl_next), static_cast<uint32_t>(sizeof(LexControl) / 4));
    LEX_HIP(hipGetLastError());
    out->stream = s;
    out->ready = ws->ready;
    out->d_packed = d_result;
    out->d_count = d_result_n;
    out->limit = lim;
    guard.armed = false;
    return RLR_OK;
}

int32_t lexical_fetch(LexPending *p, uint64_t *rows_out, float *scores_out, uint32_t *n_out)
{
    *n_out = 0;
    if (p->limit == 0 || !p->ws)
        return RLR_OK;
    LexWorkspace *ws = static_cast<LexWorkspace *>(p->ws);
    hipStream_t s = ws->stream;
    // one copy: the count sits right behind the keys; only `limit` keys can be valid
    LEX_HIP(hipMemcpyAsync(ws->h_out + kMaxLimit, ws->d_out + kMaxLimit, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    LEX_HIP(hipMemcpyAsync(ws->h_out, ws->d_out, static_cast<size_t>(p->limit) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    LEX_HIP(hipStreamSynchronize(s));
    const uint32_t n_dev = *reinterpret_cast<const uint32_t *>(ws->h_out + kMaxLimit);
    if (n_dev == kLexRetry) { // the sampled selection gave up: the caller repeats the query on the exact path
        *n_out = kLexRetry;
        return RLR_OK;
    }
    const uint32_t n = std::min<uint32_t>(n_dev, p->limit);
    for (uint32_t i = 0; i < n; ++i) {
        float sc;
        uint32_t row;
        unpack_result(ws->h_out[i], &sc, &row);
        rows_out[i] = row;
        scores_out[i] = sc;
    }
    *n_out = n;
    return RLR_OK;
}

void lexical_finish(LexPending *p, bool ok)
{
    rlr_lexical *lx = p->lx;
    if (p->ws) {
        LexWorkspace *ws = static_cast<LexWorkspace *>(p->ws);
        // the clean-up kernels behind `ready` restore the all-zero accumulators: wait for them before reuse.  After a failure
        // too: the kernels still in flight read the postings, and the shared lock that keeps a writer away from them is
        // dropped below.
        const hipError_t de = hipStreamSynchronize(ws->stream);
        const bool drained = de == hipSuccess;
        if (!drained) {
            // the stream reports an error instead of draining: make sure nothing of this device still runs before the lock
            // that protects the postings goes, leave the workspace marked dirty (it is re-zeroed before its next use) and
            // say so in the thread's error message -- the caller is on a failure path already or gets this as its first sign
            (void)hipDeviceSynchronize();
            (void)set_error(RLR_E_HIP, "lexical workspace did not drain: %s", hipGetErrorString(de));
        }
        if (ok && drained)
            ws->dirty = false;
        {
            std::lock_guard<std::mutex> lk(lx->ws_mu);
            lx->ws_free.push_back(ws);
        }
        lx->ws_cv.notify_one();
        p->ws = nullptr;
    }
    if (p->locked) {
        lx->mu.unlock_shared();
        p->locked = false;
    }
}

} // namespace rlr
