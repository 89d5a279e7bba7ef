// select.hip -- top-k nomination over the score array: two 11-bit radix-select passes
// locate the 22-bit key prefix of the k-th largest score, the guard band is subtracted,
// and every row at or above the band floor is appended to the candidate list
// (replaces the O(N log N) `scores.sort_by(..)` + `take(initial_k)` of
// /root/reference/src/rag_engine.rs:543-548; the final ordering of the handful of
// candidates happens in launch_sort_desc after the reference-order re-score).
//
// All of it is integer work on 4 bytes per row (0.13 % of the row bytes at 768-d f32);
// the score array of a 10M-row shard is 40 MB and is served from the Infinity Cache.
#include "common.h"
#include "kernels.h"
#include "select_dev.h"

#include <algorithm>
#include <cstdlib>

namespace rlr {

namespace {


__device__ inline void block_hist_flush(const uint32_t *s_hist, uint32_t *g_hist)
{
    for (int i = threadIdx.x; i < kHistBins; i += blockDim.x) {
        uint32_t c = s_hist[i];
        if (c)
            atomicAdd(&g_hist[i], c);
    }
}

// digit 1: key bits 31..21 of every score
// (all select kernels: blockIdx.y = query of a batch; strides are 0/unused for a single query)
__global__ __launch_bounds__(kSelThreads) void hist1_kernel(const float *__restrict__ scores,
                                                            uint32_t n, uint32_t *__restrict__ g_hist,
                                                            size_t score_stride, uint32_t hist_stride)
{
    __shared__ uint32_t s_hist[kHistBins];
    scores += blockIdx.y * score_stride;
    g_hist += blockIdx.y * hist_stride;
    for (int i = threadIdx.x; i < kHistBins; i += kSelThreads)
        s_hist[i] = 0;
    __syncthreads();
    const uint32_t stride = gridDim.x * kSelThreads;
    const uint32_t n4 = n / 4;
    const float4 *s4 = reinterpret_cast<const float4 *>(scores);
    for (uint32_t i = blockIdx.x * kSelThreads + threadIdx.x; i < n4; i += stride) {
        const float4 v = s4[i];
        atomicAdd(&s_hist[score_key(v.x) >> 21], 1u);
        atomicAdd(&s_hist[score_key(v.y) >> 21], 1u);
        atomicAdd(&s_hist[score_key(v.z) >> 21], 1u);
        atomicAdd(&s_hist[score_key(v.w) >> 21], 1u);
    }
    for (uint32_t i = n4 * 4 + blockIdx.x * kSelThreads + threadIdx.x; i < n; i += stride)
        atomicAdd(&s_hist[score_key(scores[i]) >> 21], 1u);
    __syncthreads();
    block_hist_flush(s_hist, g_hist);
}

// digit 2: key bits 20..10 of the scores whose digit 1 equals st->bin1
__global__ __launch_bounds__(kSelThreads) void hist2_kernel(const float *__restrict__ scores,
                                                            uint32_t n,
                                                            const SelectState *__restrict__ st,
                                                            uint32_t *__restrict__ g_hist,
                                                            size_t score_stride, uint32_t hist_stride)
{
    __shared__ uint32_t s_hist[kHistBins];
    scores += blockIdx.y * score_stride;
    g_hist += blockIdx.y * hist_stride;
    st += blockIdx.y;
    for (int i = threadIdx.x; i < kHistBins; i += kSelThreads)
        s_hist[i] = 0;
    __syncthreads();
    const uint32_t bin1 = st->bin1;
    const uint32_t stride = gridDim.x * kSelThreads;
    const uint32_t n4 = n / 4;
    const float4 *s4 = reinterpret_cast<const float4 *>(scores);
    for (uint32_t i = blockIdx.x * kSelThreads + threadIdx.x; i < n4; i += stride) {
        const float4 v = s4[i];
        const uint32_t k0 = score_key(v.x), k1 = score_key(v.y), k2 = score_key(v.z),
                       k3 = score_key(v.w);
        if ((k0 >> 21) == bin1) atomicAdd(&s_hist[(k0 >> 10) & (kHistBins - 1)], 1u);
        if ((k1 >> 21) == bin1) atomicAdd(&s_hist[(k1 >> 10) & (kHistBins - 1)], 1u);
        if ((k2 >> 21) == bin1) atomicAdd(&s_hist[(k2 >> 10) & (kHistBins - 1)], 1u);
        if ((k3 >> 21) == bin1) atomicAdd(&s_hist[(k3 >> 10) & (kHistBins - 1)], 1u);
    }
    for (uint32_t i = n4 * 4 + blockIdx.x * kSelThreads + threadIdx.x; i < n; i += stride) {
        const uint32_t k0 = score_key(scores[i]);
        if ((k0 >> 21) == bin1) atomicAdd(&s_hist[(k0 >> 10) & (kHistBins - 1)], 1u);
    }
    __syncthreads();
    block_hist_flush(s_hist, g_hist);
}

__global__ __launch_bounds__(kSelThreads) void find1_kernel(const uint32_t *__restrict__ hist1,
                                                            SelectState *st, uint32_t hist_stride)
{
    hist1 += blockIdx.x * hist_stride;
    st += blockIdx.x;
    uint32_t bin, rk;
    find_rank_bin(hist1, st->k, &bin, &rk);
    if (threadIdx.x == 0) {
        st->bin1 = bin;
        st->k2 = rk;
    }
}

__global__ __launch_bounds__(kSelThreads) void find2_kernel(const uint32_t *__restrict__ hist2,
                                                            SelectState *st, float two_eps, uint32_t hist_stride,
                                                            float *__restrict__ tau_out)
{
    hist2 += blockIdx.x * hist_stride;
    st += blockIdx.x;
    uint32_t bin, rk;
    find_rank_bin(hist2, st->k2, &bin, &rk);
    if (threadIdx.x == 0) {
        st->bin2 = bin;
        const uint32_t key_lo = band_floor_key(st->bin1, bin, two_eps);
        st->key_lo = key_lo;
        st->n_cand = 0;
        if (tau_out) // batched path: the same floor as a float for the GEMM epilogue's compare
            tau_out[blockIdx.x] = key_lo == 0 ? -__builtin_inff() : key_score(key_lo);
    }
}

// batched collect over materialised sample scores: packed (score, row) per query
__global__ __launch_bounds__(kSelThreads) void collect_packed_kernel(const float *__restrict__ scores, uint32_t n,
                                                                     SelectState *st, uint64_t *__restrict__ cand,
                                                                     size_t score_stride, uint32_t cand_stride)
{
    scores += blockIdx.y * score_stride;
    cand += static_cast<size_t>(blockIdx.y) * cand_stride;
    st += blockIdx.y;
    const uint32_t key_lo = st->key_lo;
    const uint32_t cap = st->cap;
    const uint32_t stride = gridDim.x * kSelThreads;
    for (uint32_t i = blockIdx.x * kSelThreads + threadIdx.x; i < n; i += stride) {
        const float v = scores[i];
        if (score_key(v) >= key_lo) {
            const uint32_t slot = atomicAdd(&st->n_cand, 1u);
            if (slot < cap)
                cand[slot] = pack_result(v, i);
        }
    }
}

__global__ __launch_bounds__(kSelThreads) void collect_kernel(const float *__restrict__ scores,
                                                              uint32_t n, SelectState *st,
                                                              uint32_t *__restrict__ cand)
{
    const uint32_t key_lo = st->key_lo;
    const uint32_t cap = st->cap;
    const uint32_t stride = gridDim.x * kSelThreads;
    const uint32_t n4 = n / 4;
    const float4 *s4 = reinterpret_cast<const float4 *>(scores);
    for (uint32_t i = blockIdx.x * kSelThreads + threadIdx.x; i < n4; i += stride) {
        const float4 v = s4[i];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (score_key(e[j]) >= key_lo) {
                const uint32_t slot = atomicAdd(&st->n_cand, 1u);
                if (slot < cap)
                    cand[slot] = i * 4 + j;
            }
        }
    }
    for (uint32_t i = n4 * 4 + blockIdx.x * kSelThreads + threadIdx.x; i < n; i += stride) {
        if (score_key(scores[i]) >= key_lo) {
            const uint32_t slot = atomicAdd(&st->n_cand, 1u);
            if (slot < cap)
                cand[slot] = i;
        }
    }
}

// ---- single-query pipeline: find1 folded into the digit-2 histogram, find2 into the collect --
// Every workgroup recomputes the (cheap) bin search from the 8 KB histogram in L2 instead of a
// one-workgroup kernel in between: two launches and two dependent-launch gaps fewer per query.
__global__ __launch_bounds__(kSelThreads) void hist2_find1_kernel(const float *__restrict__ scores, uint32_t n,
                                                                  const uint32_t *__restrict__ hist1,
                                                                  uint32_t *__restrict__ g_hist2, SelectState *st,
                                                                  uint32_t k, uint32_t cap)
{
    __shared__ uint32_t s_hist[kHistBins];
    uint32_t bin1, k2;
    find_rank_bin(hist1, k, &bin1, &k2);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->k = k;
        st->bin1 = bin1;
        st->k2 = k2;
        st->n_cand = 0;
        st->cap = cap;
    }
    for (int i = threadIdx.x; i < kHistBins; i += kSelThreads)
        s_hist[i] = 0;
    __syncthreads();
    const uint32_t stride = gridDim.x * kSelThreads;
    const uint32_t n4 = n / 4;
    const float4 *s4 = reinterpret_cast<const float4 *>(scores);
    for (uint32_t i = blockIdx.x * kSelThreads + threadIdx.x; i < n4; i += stride) {
        const float4 v = s4[i];
        const uint32_t k0 = score_key(v.x), k1 = score_key(v.y), k2_ = score_key(v.z), k3 = score_key(v.w);
        if ((k0 >> 21) == bin1) atomicAdd(&s_hist[(k0 >> 10) & (kHistBins - 1)], 1u);
        if ((k1 >> 21) == bin1) atomicAdd(&s_hist[(k1 >> 10) & (kHistBins - 1)], 1u);
        if ((k2_ >> 21) == bin1) atomicAdd(&s_hist[(k2_ >> 10) & (kHistBins - 1)], 1u);
        if ((k3 >> 21) == bin1) atomicAdd(&s_hist[(k3 >> 10) & (kHistBins - 1)], 1u);
    }
    for (uint32_t i = n4 * 4 + blockIdx.x * kSelThreads + threadIdx.x; i < n; i += stride) {
        const uint32_t k0 = score_key(scores[i]);
        if ((k0 >> 21) == bin1) atomicAdd(&s_hist[(k0 >> 10) & (kHistBins - 1)], 1u);
    }
    __syncthreads();
    block_hist_flush(s_hist, g_hist2);
}

__global__ __launch_bounds__(kSelThreads) void collect_find2_kernel(const float *__restrict__ scores, uint32_t n,
                                                                    const uint32_t *__restrict__ hist2,
                                                                    SelectState *st, float two_eps,
                                                                    uint32_t *__restrict__ cand)
{
    uint32_t bin2, rk;
    find_rank_bin(hist2, st->k2, &bin2, &rk);
    const uint32_t key_lo = band_floor_key(st->bin1, bin2, two_eps);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->bin2 = bin2;
        st->key_lo = key_lo;
    }
    const uint32_t cap = st->cap;
    const uint32_t stride = gridDim.x * kSelThreads;
    const uint32_t n4 = n / 4;
    const float4 *s4 = reinterpret_cast<const float4 *>(scores);
    for (uint32_t i = blockIdx.x * kSelThreads + threadIdx.x; i < n4; i += stride) {
        const float4 v = s4[i];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (score_key(e[j]) >= key_lo) {
                const uint32_t slot = atomicAdd(&st->n_cand, 1u);
                if (slot < cap)
                    cand[slot] = i * 4 + j;
            }
        }
    }
    for (uint32_t i = n4 * 4 + blockIdx.x * kSelThreads + threadIdx.x; i < n; i += stride) {
        if (score_key(scores[i]) >= key_lo) {
            const uint32_t slot = atomicAdd(&st->n_cand, 1u);
            if (slot < cap)
                cand[slot] = i;
        }
    }
}

// ---- batched select, fused: one workgroup per query ------------------------------------------------------------------
// hist1 -> find1 -> hist2 -> find2 -> collect were five launches (three of them a pass of every workgroup over every query's
// sample scores with global-memory histograms and candidate counters).  A sample is 65 k - 150 k scores per query, 0.3 - 0.6 MB:
// one workgroup of 1024 threads reads it three times out of L2 with its histograms and its candidate counter in LDS and
// leaves exactly what the five kernels left -- SelectState (bin1, k2, bin2, key_lo, n_cand), tau, and the candidates at or
// above the floor as packed (score, row) words.
__global__ __launch_bounds__(1024) void batch_select_fused_kernel(const float *__restrict__ scores, uint32_t n, size_t score_stride,
                                                                  SelectState *__restrict__ st, float two_eps,
                                                                  float *__restrict__ tau_out, uint64_t *__restrict__ cand,
                                                                  uint32_t cand_stride)
{
    __shared__ uint32_t s_hist[kHistBins];
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_res[2];
    __shared__ uint32_t s_count;
    scores += blockIdx.x * score_stride;
    cand += static_cast<size_t>(blockIdx.x) * cand_stride;
    st += blockIdx.x;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t n4 = n / 4;
    const float4 *s4 = reinterpret_cast<const float4 *>(scores);
    // rank-th largest key's bin in s_hist (2048 bins, two per thread), from the top: wave suffix scan + the 16 wave totals
    auto find_bin = [&](uint32_t rank, uint32_t *bin_out, uint32_t *rank_in_bin) {
        const uint32_t lo = s_hist[2 * t], hi = s_hist[2 * t + 1];
        const uint32_t sum = lo + hi;
        uint32_t suf = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_down(suf, off);
            if (lane + off < 64)
                suf += v;
        }
        if (lane == 0)
            s_wave[wave] = suf;
        if (t == 0) {
            s_res[0] = 0;
            s_res[1] = 1;
        }
        __syncthreads();
        uint32_t above = suf - sum; // keys in bins above this thread's two
        for (uint32_t w = wave + 1; w < 16; ++w)
            above += s_wave[w];
        if (above < rank && rank <= above + hi) {
            s_res[0] = 2 * t + 1;
            s_res[1] = rank - above;
        } else if (above + hi < rank && rank <= above + hi + lo) {
            s_res[0] = 2 * t;
            s_res[1] = rank - above - hi;
        }
        __syncthreads();
        *bin_out = s_res[0];
        *rank_in_bin = s_res[1];
        __syncthreads();
    };
    // digit 1
    s_hist[2 * t] = 0;
    s_hist[2 * t + 1] = 0;
    if (t == 0)
        s_count = 0;
    __syncthreads();
    for (uint32_t i = t; i < n4; i += 1024) {
        const float4 v = s4[i];
        atomicAdd(&s_hist[score_key(v.x) >> 21], 1u);
        atomicAdd(&s_hist[score_key(v.y) >> 21], 1u);
        atomicAdd(&s_hist[score_key(v.z) >> 21], 1u);
        atomicAdd(&s_hist[score_key(v.w) >> 21], 1u);
    }
    for (uint32_t i = n4 * 4 + t; i < n; i += 1024)
        atomicAdd(&s_hist[score_key(scores[i]) >> 21], 1u);
    __syncthreads();
    uint32_t bin1, k2;
    find_bin(st->k, &bin1, &k2);
    // digit 2 of the scores inside bin1
    s_hist[2 * t] = 0;
    s_hist[2 * t + 1] = 0;
    __syncthreads();
    for (uint32_t i = t; i < n4; i += 1024) {
        const float4 v = s4[i];
        const uint32_t k0 = score_key(v.x), k1 = score_key(v.y), k2_ = score_key(v.z), k3 = score_key(v.w);
        if ((k0 >> 21) == bin1) atomicAdd(&s_hist[(k0 >> 10) & (kHistBins - 1)], 1u);
        if ((k1 >> 21) == bin1) atomicAdd(&s_hist[(k1 >> 10) & (kHistBins - 1)], 1u);
        if ((k2_ >> 21) == bin1) atomicAdd(&s_hist[(k2_ >> 10) & (kHistBins - 1)], 1u);
        if ((k3 >> 21) == bin1) atomicAdd(&s_hist[(k3 >> 10) & (kHistBins - 1)], 1u);
    }
    for (uint32_t i = n4 * 4 + t; i < n; i += 1024) {
        const uint32_t k0 = score_key(scores[i]);
        if ((k0 >> 21) == bin1) atomicAdd(&s_hist[(k0 >> 10) & (kHistBins - 1)], 1u);
    }
    __syncthreads();
    uint32_t bin2, rk;
    find_bin(k2, &bin2, &rk);
    const uint32_t key_lo = band_floor_key(bin1, bin2, two_eps);
    // collect: everything at or above the floor
    const uint32_t cap = st->cap;
    for (uint32_t i = t; i < n4; i += 1024) {
        const float4 v = s4[i];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (score_key(e[j]) >= key_lo) {
                const uint32_t slot = atomicAdd(&s_count, 1u);
                if (slot < cap)
                    cand[slot] = pack_result(e[j], i * 4 + j);
            }
    }
    for (uint32_t i = n4 * 4 + t; i < n; i += 1024) {
        const float v = scores[i];
        if (score_key(v) >= key_lo) {
            const uint32_t slot = atomicAdd(&s_count, 1u);
            if (slot < cap)
                cand[slot] = pack_result(v, i);
        }
    }
    __syncthreads();
    if (t == 0) {
        st->bin1 = bin1;
        st->k2 = k2;
        st->bin2 = bin2;
        st->key_lo = key_lo;
        st->n_cand = s_count;
        if (tau_out)
            tau_out[blockIdx.x] = key_lo == 0 ? -__builtin_inff() : key_score(key_lo);
    }
}

// ---- descending sort of packed u64 ------------------------------------------------
constexpr int kSortLds = 4096; // entries sorted inside one workgroup's LDS (32 KB)

__global__ __launch_bounds__(1024) void sort_lds_kernel(uint64_t *data, uint32_t n_pad)
{
    __shared__ uint64_t s[kSortLds];
    for (uint32_t i = threadIdx.x; i < n_pad; i += 1024)
        s[i] = data[i];
    __syncthreads();
    for (uint32_t k = 2; k <= n_pad; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n_pad; i += 1024) {
                const uint32_t ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = s[i], b = s[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) {
                        s[i] = b;
                        s[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < n_pad; i += 1024)
        data[i] = s[i];
}

// one (k, j) step of a global-memory bitonic network (band-overflow path only)
__global__ __launch_bounds__(256) void sort_global_step(uint64_t *data, uint32_t n_pad, uint32_t k,
                                                        uint32_t j)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pad)
        return;
    const uint32_t ixj = i ^ j;
    if (ixj > i) {
        const uint64_t a = data[i], b = data[ixj];
        const bool desc = (i & k) == 0;
        if (desc ? (a < b) : (a > b)) {
            data[i] = b;
            data[ixj] = a;
        }
    }
}

uint32_t sel_blocks(uint32_t n, int n_cu)
{
    uint32_t want = (n / 4 + kSelThreads - 1) / kSelThreads;
    uint32_t cap = static_cast<uint32_t>(n_cu) * 8;
    if (want > cap) want = cap;
    return want ? want : 1;
}

} // namespace

// Batched two-pass select over q_count score rows of length n (stride score_stride):
// leaves key_lo / tau per query and the sample's candidates in cand[q][...].
hipError_t launch_batch_select(const float *scores, uint32_t n, size_t score_stride, uint32_t q_count,
                               uint32_t *hist /* q_count x 2 x kHistBins, zeroed */, SelectState *st,
                               float two_eps, float *tau_out, uint64_t *cand, uint32_t cand_stride, int n_cu,
                               hipStream_t s)
{
    // one workgroup per query while the queries alone fill the chip and a sample stays L2-sized: three passes over its own
    // 4 n bytes with every counter in LDS (RLR_BATCH_SELECT_SPLIT=1: the five-launch form)
    static const bool split = getenv("RLR_BATCH_SELECT_SPLIT") != nullptr;
    if (!split && q_count >= 64 && n <= (1u << 20) && (score_stride % 4) == 0) {
        hipLaunchKernelGGL(batch_select_fused_kernel, dim3(q_count), dim3(1024), 0, s, scores, n, score_stride, st, two_eps, tau_out,
                           cand, cand_stride);
        return hipGetLastError();
    }
    uint32_t bx = (n / 4 + kSelThreads - 1) / kSelThreads;
    const uint32_t bx_cap = std::max<uint32_t>(1u, static_cast<uint32_t>(n_cu) * 8 / std::max(q_count, 1u));
    bx = std::max(1u, std::min(bx, bx_cap));
    const uint32_t hs = 2 * kHistBins;
    hipLaunchKernelGGL(hist1_kernel, dim3(bx, q_count), dim3(kSelThreads), 0, s, scores, n, hist, score_stride, hs);
    hipLaunchKernelGGL(find1_kernel, dim3(q_count), dim3(kSelThreads), 0, s, hist, st, hs);
    hipLaunchKernelGGL(hist2_kernel, dim3(bx, q_count), dim3(kSelThreads), 0, s, scores, n, st, hist + kHistBins,
                       score_stride, hs);
    hipLaunchKernelGGL(find2_kernel, dim3(q_count), dim3(kSelThreads), 0, s, hist + kHistBins, st, two_eps, hs, tau_out);
    hipLaunchKernelGGL(collect_packed_kernel, dim3(bx, q_count), dim3(kSelThreads), 0, s, scores, n, st, cand,
                       score_stride, cand_stride);
    return hipGetLastError();
}

hipError_t launch_collect(const float *scores, uint32_t n, SelectState *st, uint32_t *cand,
                          int n_cu, hipStream_t s)
{
    hipLaunchKernelGGL(collect_kernel, dim3(sel_blocks(n, n_cu)), dim3(kSelThreads), 0, s, scores, n,
                       st, cand);
    return hipGetLastError();
}

hipError_t launch_hist2_find1(const float *scores, uint32_t n, const uint32_t *hist1, uint32_t *hist2,
                              SelectState *st, uint32_t k, uint32_t cap, int n_cu, hipStream_t s)
{
    hipLaunchKernelGGL(hist2_find1_kernel, dim3(sel_blocks(n, n_cu)), dim3(kSelThreads), 0, s, scores, n, hist1, hist2,
                       st, k, cap);
    return hipGetLastError();
}

hipError_t launch_collect_find2(const float *scores, uint32_t n, const uint32_t *hist2, SelectState *st,
                                float two_eps, uint32_t *cand, int n_cu, hipStream_t s)
{
    hipLaunchKernelGGL(collect_find2_kernel, dim3(sel_blocks(n, n_cu)), dim3(kSelThreads), 0, s, scores, n, hist2, st,
                       two_eps, cand);
    return hipGetLastError();
}

hipError_t launch_sort_desc(uint64_t *packed, uint32_t n_pad, hipStream_t s)
{
    if (n_pad <= 1)
        return hipSuccess;
    if (n_pad <= kSortLds) {
        hipLaunchKernelGGL(sort_lds_kernel, dim3(1), dim3(1024), 0, s, packed, n_pad);
        return hipGetLastError();
    }
    const uint32_t blocks = (n_pad + 255) / 256;
    for (uint32_t k = 2; k <= n_pad; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1)
            hipLaunchKernelGGL(sort_global_step, dim3(blocks), dim3(256), 0, s, packed, n_pad, k, j);
    return hipGetLastError();
}

} // namespace rlr
