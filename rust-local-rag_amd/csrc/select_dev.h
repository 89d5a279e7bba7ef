// select_dev.h -- device helpers of the radix select shared by select.hip (the stand-alone stages) and tail.hip (the
// fused select -> re-score -> sort tail): the bin search over a 2048-bin histogram and the guard-band floor.
#pragma once

#include "common.h"
#include "kernels.h"

namespace rlr {

constexpr int kSelThreads = 256;

// One workgroup: find the bin holding the `rank`-th largest key (rank is 1-based) by a suffix
// sum from the top bin: 8 bins per thread, wavefront suffix scan with __shfl_down, the four wave
// totals combined through LDS.  Returns (bin, rank inside the bin) to every thread.
__device__ inline void find_rank_bin(const uint32_t *__restrict__ hist, uint32_t rank,
                                     uint32_t *bin_out, uint32_t *rank_in_bin, uint32_t *count_out = nullptr)
{
    constexpr int PER = kHistBins / kSelThreads; // 8 bins per thread
    __shared__ uint32_t s_wave[kSelThreads / 64];
    __shared__ uint32_t s_res[3];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    uint32_t loc[PER];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        loc[i] = hist[t * PER + i];
        sum += loc[i];
    }
    // inclusive suffix sum inside the wave: suf = sum over lanes >= lane
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
        s_res[2] = 0;
    }
    __syncthreads();
    uint32_t above_waves = 0; // keys in bins owned by higher waves
    for (int w = wave + 1; w < kSelThreads / 64; ++w)
        above_waves += s_wave[w];
    // above = number of keys in bins above this thread's highest bin
    uint32_t above = above_waves + suf - sum;
#pragma unroll
    for (int i = PER - 1; i >= 0; --i) {
        const uint32_t with = above + loc[i];
        if (above < rank && rank <= with) {
            s_res[0] = t * PER + i;
            s_res[1] = rank - above;
            s_res[2] = loc[i];
        }
        above = with;
    }
    __syncthreads();
    *bin_out = s_res[0];
    *rank_in_bin = s_res[1];
    if (count_out)
        *count_out = s_res[2]; // keys in that bin (0 when the histogram holds fewer than `rank` keys)
}

__device__ inline uint32_t band_floor_key(uint32_t bin1, uint32_t bin2, float two_eps)
{
    // lower edge of the 22-bit prefix bin that holds the k-th largest score, minus the band
    const uint32_t key_floor = (bin1 << 21) | (bin2 << 10);
    if (key_floor == 0)
        return 0;
    const float lo = key_score(key_floor) - two_eps; // -inf stays -inf
    const uint32_t key_lo = score_key(lo);
    return key_lo > key_floor ? key_floor : key_lo;
}

} // namespace rlr
