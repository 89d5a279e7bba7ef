// lds_select.h -- selection helpers that work on a candidate list held in LDS (one workgroup).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rlr {

// Number of keys in s[0, n) (LDS) greater than `mine`: the position of `mine` in the descending order of unique keys (the
// rank sorts of sort_emit.h / pool_prepare.h / the merge kernel).  Eight independent 8-byte reads in flight per step: written
// as a plain `for (j) rank += s[j] > mine` loop every iteration waited for its own LDS read (~100 cycles), and the
// few-hundred-candidate sorts cost 5-6 us each on a workgroup whose other waves have nothing to do meanwhile.
__device__ inline uint32_t lds_rank_desc(const uint64_t *s, uint32_t n, uint64_t mine)
{
    uint32_t rank = 0, j = 0;
    for (; j + 8 <= n; j += 8) {
        uint64_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = s[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            rank += v[u] > mine;
    }
    for (; j < n; ++j)
        rank += s[j] > mine;
    return rank;
}

// The bin holding the rank-th key counted from the top of a histogram of up to 2048 bins in LDS (entries [nb, 2048) zero), by
// all 1024 threads of the workgroup: thread t owns bins 2t and 2t + 1 (one conflict-free read), a wavefront suffix
// scan, the 16 wave totals through LDS.  sel[0] = bin, sel[1] = rank inside the bin (1-based), sel[2] = the bin's count;
// valid after the call's last barrier.  (It was one wavefront walking 32 bins per lane at a 32-word stride -- every read a
// 32-way bank conflict -- twice: ~1.5 us of each radix pass.)
__device__ inline void lds_find_rank_bin_1024(const uint32_t *s_hist, uint32_t rank, uint32_t *s_sel)
{
    __shared__ uint32_t s_wave_tot[16];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t lo = s_hist[2 * t], hi = s_hist[2 * t + 1], sum = lo + hi; // (one ds_read2_b32)
    uint32_t suf = sum; // sum over this wave's lanes >= lane
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_down(suf, off);
        if (lane + off < 64)
            suf += v;
    }
    if (lane == 0)
        s_wave_tot[wave] = suf;
    __syncthreads();
    uint32_t above = suf - sum; // keys in bins above this thread's two
    for (uint32_t w = wave + 1; w < 16; ++w)
        above += s_wave_tot[w];
    if (above < rank && rank <= above + hi) {
        s_sel[0] = 2 * t + 1;
        s_sel[1] = rank - above;
        s_sel[2] = hi;
    } else if (above + hi < rank && rank <= above + hi + lo) {
        s_sel[0] = 2 * t;
        s_sel[1] = rank - above - hi;
        s_sel[2] = lo;
    }
    __syncthreads();
}

// k-th largest 32-bit key among the packed candidates s_c[0, n) (key = high word), by three radix passes over LDS
// (11 + 11 + 10 bits; histogram by LDS atomics, the bin search by one wavefront).  All threads must call it; the result
// is returned to every thread.  Replaces a full bitonic sort of the list (91 barrier-separated stages at 8192 entries:
// 179 us per batch of 256 queries) where only the k-th score and the set above a threshold are needed.
__device__ inline uint32_t lds_kth_key(const uint64_t *s_c, uint32_t n, uint32_t k, uint32_t *s_hist, uint32_t *s_sel,
                                       uint32_t nthreads)
{
    uint32_t prefix = 0, mask = 0, rank = k; // rank: 1-based position, counted from the largest, inside the current bin
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
        const uint32_t shift = pass == 0 ? 21u : pass == 1 ? 10u : 0u;
        const uint32_t nb = pass == 2 ? 1024u : 2048u;
        for (uint32_t i = threadIdx.x; i < (nthreads == 1024 ? 2048u : nb); i += nthreads)
            s_hist[i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += nthreads) {
            const uint32_t key = static_cast<uint32_t>(s_c[i] >> 32);
            if ((key & mask) == prefix)
                atomicAdd(&s_hist[(key >> shift) & (nb - 1)], 1u);
        }
        __syncthreads();
        if (nthreads == 1024) {
            lds_find_rank_bin_1024(s_hist, rank, s_sel);
        } else if (threadIdx.x < 64) {
            // lane l owns bins [l * W, (l + 1) * W); suffix sums over the lanes from the top, then inside the lane
            const uint32_t W = nb / 64, lane = threadIdx.x;
            uint32_t mine = 0;
            for (uint32_t b = 0; b < W; ++b)
                mine += s_hist[lane * W + b];
            uint32_t incl = mine; // sum over lanes >= lane
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_down(incl, d);
                if (lane + d < 64)
                    incl += o;
            }
            const uint32_t above = incl - mine; // keys in bins of higher lanes
            if (above < rank && rank <= incl) {
                uint32_t acc = above;
                for (int b = static_cast<int>(W) - 1; b >= 0; --b) {
                    const uint32_t h = s_hist[lane * W + b];
                    if (rank <= acc + h) {
                        s_sel[0] = lane * W + b;
                        s_sel[1] = rank - acc;
                        break;
                    }
                    acc += h;
                }
            }
        }
        if (nthreads != 1024)
            __syncthreads();
        prefix |= s_sel[0] << shift;
        mask |= (nb - 1) << shift;
        rank = s_sel[1];
        __syncthreads();
    }
    return prefix;
}

// k-th largest full 64-bit key of s_c[0, n) (keys are unique where the caller needs exactly k winners): radix passes over
// LDS with 11-bit digits, same structure as lds_kth_key.  All threads must call it (n >= 1).
//
// `slack`: the passes stop as soon as the bin holding the k-th key has at most `slack` keys ranked BELOW it (bin
// population - rank inside the bin <= slack); the bin's lower edge is returned then, so between k and k + slack keys are
// >= the result.  slack = 0 still yields exactly k keys (it stops when the whole bin belongs to the top k -- with unique
// keys usually after two or three passes); a threshold that may admit a few more uses a larger slack.
//
// Only bits that DIFFER between keys get a pass: one AND / OR reduction over the keys in front of the first pass finds the
// bits every key shares (the sign and most of the exponent of scores from a narrow range; the ones above the highest row
// number in the inverted row word), and every digit starts at the highest varying bit that is still undecided.  With the
// digits on a fixed grid from bit 63 the first pass of a BM25 selection put 8192 keys into a dozen bins -- thousands of
// LDS atomics on one address, and a pass that decided almost nothing.  (`row_bits` told the fixed grid where the row
// word's constant ones end; the reduction finds that by itself and the argument is ignored.)
__device__ inline uint64_t lds_kth_key64(const uint64_t *s_c, uint32_t n, uint32_t k, uint32_t *s_hist, uint32_t *s_sel,
                                         uint32_t nthreads, uint32_t slack = 0, uint32_t row_bits = 32)
{
    (void)row_bits;
    uint64_t all_and = ~0ull, all_or = 0ull;
    for (uint32_t i = threadIdx.x; i < n; i += nthreads) {
        const uint64_t key = s_c[i];
        all_and &= key;
        all_or |= key;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        all_and &= __shfl_xor(all_and, off);
        all_or |= __shfl_xor(all_or, off);
    }
    uint64_t *s_red = reinterpret_cast<uint64_t *>(s_hist); // (the histogram is cleared in front of every pass)
    const uint32_t n_waves = nthreads / 64, wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_red[2 * wave] = all_and;
        s_red[2 * wave + 1] = all_or;
    }
    __syncthreads();
    for (uint32_t w = 0; w < n_waves; ++w) {
        all_and &= s_red[2 * w];
        all_or |= s_red[2 * w + 1];
    }
    __syncthreads();
    const uint64_t diff = all_and ^ all_or; // the bits that are not the same in every key
    uint64_t prefix = all_and & ~diff, mask = ~diff;
    uint32_t rank = k;
    uint32_t next_hi = 64; // bits [next_hi, 64) are decided
#pragma unroll 1
    for (int pass = 0; pass < 8; ++pass) {
        const uint64_t open = next_hi >= 64 ? diff : diff & ((1ull << next_hi) - 1ull);
        if (open == 0)
            break; // the keys still in the race agree in every remaining bit
        const uint32_t top = 64u - static_cast<uint32_t>(__clzll(static_cast<long long>(open))); // one past the highest open bit
        const uint32_t bits = top < 11u ? top : 11u;
        const uint32_t shift = top - bits, nb = 1u << bits;
        for (uint32_t i = threadIdx.x; i < (nthreads == 1024 ? 2048u : nb); i += nthreads)
            s_hist[i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += nthreads) {
            const uint64_t key = s_c[i];
            if ((key & mask) == prefix)
                atomicAdd(&s_hist[static_cast<uint32_t>(key >> shift) & (nb - 1)], 1u);
        }
        __syncthreads();
        if (nthreads == 1024) {
            lds_find_rank_bin_1024(s_hist, rank, s_sel);
        } else if (threadIdx.x < 64) {
            // lane l owns bins [l * W, (l + 1) * W) (W >= 1: digits narrower than 6 bits leave the upper lanes without bins)
            const uint32_t W = nb >= 64 ? nb / 64 : 1u, lane = threadIdx.x;
            const bool owns = lane * W < nb;
            uint32_t mine = 0;
            if (owns)
                for (uint32_t b = 0; b < W; ++b)
                    mine += s_hist[lane * W + b];
            uint32_t incl = mine; // sum over lanes >= lane
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_down(incl, d);
                if (lane + d < 64)
                    incl += o;
            }
            const uint32_t above = incl - mine;
            if (owns && above < rank && rank <= incl) {
                uint32_t acc = above;
                for (int b = static_cast<int>(W) - 1; b >= 0; --b) {
                    const uint32_t h = s_hist[lane * W + b];
                    if (rank <= acc + h) {
                        s_sel[0] = lane * W + b;
                        s_sel[1] = rank - acc;
                        s_sel[2] = h;
                        break;
                    }
                    acc += h;
                }
            }
        }
        if (nthreads != 1024)
            __syncthreads();
        prefix |= static_cast<uint64_t>(s_sel[0]) << shift;
        mask |= static_cast<uint64_t>(nb - 1) << shift;
        rank = s_sel[1];
        const bool done = s_sel[2] - rank <= slack; // (uniform) everything the bin holds below the k-th key is tolerated
        __syncthreads();
        next_hi = shift;
        if (done || shift == 0)
            break;
    }
    return prefix;
}

} // namespace rlr
