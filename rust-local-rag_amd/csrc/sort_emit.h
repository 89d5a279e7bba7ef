// sort_emit.h -- the last step of a single-query search: one workgroup of 1024 threads orders the exactly re-scored
// candidates (packed (score key, ~row) words, <= 4096 of them) and writes the best k.  Replaces the tail of
// `scores.sort_by(..)` + `take(..)`, /root/reference/src/rag_engine.rs:543-548, for the handful of rows that survived
// the nomination.  Shared by index.hip (the stand-alone launch) and tail.hip (the fused select tail).
#pragma once

#include "common.h"
#include "lds_select.h"

namespace rlr {

// `s`: 4096 u64 of LDS, `s_hist`: 2048 u32 of LDS (only touched on the `unordered` path).  Called by all 1024 threads.
// unordered: more than 1024 candidates are not sorted (a 2048-entry bitonic network: 22 us) -- the k best are found by a
// radix select of the k-th key and written in any order (valid entries first, zeros behind, as in the sorted form).
__device__ inline void sort_emit_body(const uint64_t *__restrict__ packed, uint32_t n_raw, uint32_t cap,
                                      uint64_t *__restrict__ out, uint32_t k, uint64_t *__restrict__ meta, bool unordered,
                                      uint64_t *s, uint32_t *s_hist)
{
    if (threadIdx.x == 0 && meta)
        *meta = n_raw; // travels to the host with the results: one D2H per call
    if (n_raw > cap || n_raw > 4096) {
        // band overflow: the host re-runs this query on the large-candidate path.  The all-ones word marks
        // the slot invalid for consumers that read it before the host has looked (the sharded merge).
        if (threadIdx.x == 0)
            out[0] = ~0ull;
        return;
    }
    if (n_raw <= 1024) {
        // rank sort: keys are unique (the row number is part of the key), so the number of larger
        // keys is the output position -- one pass, two barriers, instead of a log^2 network.
        if (threadIdx.x < n_raw)
            s[threadIdx.x] = packed[threadIdx.x];
        __syncthreads();
        if (threadIdx.x < n_raw) {
            const uint64_t mine = s[threadIdx.x];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n_raw; ++j)
                rank += s[j] > mine;
            if (rank < k)
                out[rank] = mine;
        }
        for (uint32_t i = n_raw + threadIdx.x; i < k; i += 1024)
            out[i] = 0ull;
        return;
    }
    if (unordered) {
        __shared__ uint32_t s_pick[3];
        __shared__ uint32_t s_n;
        for (uint32_t i = threadIdx.x; i < n_raw; i += 1024)
            s[i] = packed[i];
        if (threadIdx.x == 0)
            s_n = 0;
        __syncthreads();
        uint64_t kth = 0;
        if (n_raw > k)
            kth = lds_kth_key64(s, n_raw, k, s_hist, s_pick, 1024); // unique keys: exactly k of them are >= kth
        for (uint32_t i = threadIdx.x; i < n_raw; i += 1024) {
            const uint64_t v = s[i];
            if (v >= kth)
                out[atomicAdd(&s_n, 1u)] = v;
        }
        for (uint32_t i = min(n_raw, k) + threadIdx.x; i < k; i += 1024)
            out[i] = 0ull;
        return;
    }
    uint32_t n_pad = 1;
    while (n_pad < n_raw)
        n_pad <<= 1;
    for (uint32_t i = threadIdx.x; i < n_pad; i += 1024)
        s[i] = i < n_raw ? packed[i] : 0ull;
    __syncthreads();
    for (uint32_t kk = 2; kk <= n_pad; kk <<= 1) {
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n_pad; i += 1024) {
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
    for (uint32_t i = threadIdx.x; i < k; i += 1024)
        out[i] = i < n_raw ? s[i] : 0ull;
}

} // namespace rlr
