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
// COHERENT: the candidates were written by other workgroups of the SAME launch (agent-scope stores): read them with
// agent-scope loads.
template <bool COHERENT = false>
__device__ inline uint64_t load_candidate(const uint64_t *p)
{
    if constexpr (COHERENT)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        return *p;
}

// Position-dependent 32-bit checksum over emitted 64-bit words: sum of result_chk_term(word, position).  The device
// accumulates it while it writes the words, the host recomputes it over what it finds in its (pinned) memory: equality
// means every word has arrived.  A zero word contributes nothing (the padding behind fewer than k results).
__host__ __device__ inline uint32_t result_chk_term(uint64_t w, uint32_t i)
{
    const uint32_t lo = static_cast<uint32_t>(w), hi = static_cast<uint32_t>(w >> 32);
    return (lo * 0x9E3779B1u + hi * 0x85EBCA77u + (lo ^ hi)) * (2u * i + 1u);
}

// `s_chk`: one LDS word, zero on entry, the checksum of out[0, k) on return (valid after the caller's next barrier).
// `pre`: packed[threadIdx.x] already loaded by the caller (issued before it knew n_raw, to overlap the two round trips), or null.
template <bool COHERENT = false>
__device__ inline void sort_emit_rows(const uint64_t *packed, uint32_t n_raw, uint32_t cap, uint64_t *__restrict__ out,
                                      uint32_t k, bool unordered, uint64_t *s, uint32_t *s_hist, uint32_t *s_chk,
                                      const uint64_t *pre = nullptr)
{
    if (n_raw > cap || n_raw > 4096) {
        // band overflow: the host re-runs this query on the large-candidate path.  The all-ones word marks
        // the slot invalid for consumers that read it before the host has looked (the sharded merge).
        if (threadIdx.x == 0) {
            out[0] = ~0ull;
            atomicAdd(s_chk, result_chk_term(~0ull, 0));
        }
        return;
    }
    if (n_raw <= 1024) {
        // rank sort: keys are unique (the row number is part of the key), so the number of larger
        // keys is the output position -- one pass, two barriers, instead of a log^2 network.
        if (threadIdx.x < n_raw)
            s[threadIdx.x] = pre ? *pre : load_candidate<COHERENT>(packed + threadIdx.x);
        __syncthreads();
        if (threadIdx.x < n_raw) {
            const uint64_t mine = s[threadIdx.x];
            const uint32_t rank = lds_rank_desc(s, n_raw, mine);
            if (rank < k) {
                out[rank] = mine;
                atomicAdd(s_chk, result_chk_term(mine, rank));
            }
        }
        for (uint32_t i = n_raw + threadIdx.x; i < k; i += 1024)
            out[i] = 0ull;
        return;
    }
    if (unordered) {
        __shared__ uint32_t s_pick[3];
        __shared__ uint32_t s_n;
        for (uint32_t i = threadIdx.x; i < n_raw; i += 1024)
            s[i] = load_candidate<COHERENT>(packed + i);
        if (threadIdx.x == 0)
            s_n = 0;
        __syncthreads();
        uint64_t kth = 0;
        if (n_raw > k)
            kth = lds_kth_key64(s, n_raw, k, s_hist, s_pick, 1024); // unique keys: exactly k of them are >= kth
        for (uint32_t i = threadIdx.x; i < n_raw; i += 1024) {
            const uint64_t v = s[i];
            if (v >= kth) {
                const uint32_t at = atomicAdd(&s_n, 1u);
                out[at] = v;
                atomicAdd(s_chk, result_chk_term(v, at));
            }
        }
        for (uint32_t i = min(n_raw, k) + threadIdx.x; i < k; i += 1024)
            out[i] = 0ull;
        return;
    }
    if (k <= 1024) {
        // more than 1024 candidates, at most 1024 wanted, in order: select the k-th key by radix passes over LDS, move the k
        // winners (keys are unique: exactly k of them are >= it) into the histogram's space and rank-sort those -- instead
        // of a bitonic network over all of them (4096 entries: 78 barrier-separated stages)
        __shared__ uint32_t s_pick2[3];
        __shared__ uint32_t s_n2;
        for (uint32_t i = threadIdx.x; i < n_raw; i += 1024)
            s[i] = load_candidate<COHERENT>(packed + i);
        if (threadIdx.x == 0)
            s_n2 = 0;
        __syncthreads();
        const uint64_t kth = lds_kth_key64(s, n_raw, k, s_hist, s_pick2, 1024);
        uint64_t *win = reinterpret_cast<uint64_t *>(s_hist); // 2048 u32 = 1024 u64; the select is done with it
        for (uint32_t i = threadIdx.x; i < n_raw; i += 1024) {
            const uint64_t v = s[i];
            if (v >= kth)
                win[atomicAdd(&s_n2, 1u)] = v;
        }
        __syncthreads();
        const uint32_t m = s_n2; // == k
        if (threadIdx.x < m) {
            const uint64_t mine = win[threadIdx.x];
            const uint32_t rank = lds_rank_desc(win, m, mine);
            out[rank] = mine;
            atomicAdd(s_chk, result_chk_term(mine, rank));
        }
        return;
    }
    uint32_t n_pad = 1;
    while (n_pad < n_raw)
        n_pad <<= 1;
    for (uint32_t i = threadIdx.x; i < n_pad; i += 1024)
        s[i] = i < n_raw ? load_candidate<COHERENT>(packed + i) : 0ull;
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
    for (uint32_t i = threadIdx.x; i < k; i += 1024) {
        const uint64_t v = i < n_raw ? s[i] : 0ull;
        out[i] = v;
        if (v)
            atomicAdd(s_chk, result_chk_term(v, i));
    }
}

// The rows, then -- behind a system-scope fence and a barrier -- one word in *meta: the candidate count in the low half,
// the checksum of the k emitted words in the high half.  A host that pre-set *meta (pinned memory) to kMetaPending can poll
// it instead of waiting for the stream's completion signal (index.hip: wait_results): when the word has arrived AND the
// checksum of what lies in its result buffer matches, the k results are complete.  (The fence alone does not order the
// word behind the results for a reader on the other side of PCIe: one query in ~5000 returned with three of the previous
// query's rows still in the buffer -- the checksum closes that window end to end.)
constexpr uint64_t kMetaPending = 0xFFFFFFFFFFFFFFFEull;

template <bool COHERENT = false>
__device__ inline void sort_emit_body(const uint64_t *packed, uint32_t n_raw, uint32_t cap,
                                      uint64_t *__restrict__ out, uint32_t k, uint64_t *__restrict__ meta, bool unordered,
                                      uint64_t *s, uint32_t *s_hist, const uint64_t *pre = nullptr)
{
    __shared__ uint32_t s_chk;
    if (threadIdx.x == 0)
        s_chk = 0;
    __syncthreads();
    sort_emit_rows<COHERENT>(packed, n_raw, cap, out, k, unordered, s, s_hist, &s_chk, pre);
    if (!meta)
        return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t m = (static_cast<uint64_t>(s_chk) << 32) | n_raw;
        if (m == kMetaPending)
            m ^= 1ull << 32; // (never the pending pattern; the host accepts either checksum for a count of 0xFFFFFFFE)
        *meta = m;
    }
}

// The result block of the fused search -> MMR paths in pinned host memory: [row | cos | combined | lexical] x k_cap, then
// n, status, checksum, done.  `done` (pre-set to kBlockPending by a host that wants to poll) is written last; the
// checksum covers the 4 x n value words plus n and status, with the same term as above.
constexpr uint32_t kBlockPending = 0xFFFFFFFFu, kBlockDone = 1u;

__host__ __device__ inline uint32_t block_chk_tail(uint32_t chk_values, uint32_t n, uint32_t status, uint32_t k_cap)
{
    return chk_values + result_chk_term(n, 4 * k_cap) + result_chk_term(status, 4 * k_cap + 1);
}

} // namespace rlr
