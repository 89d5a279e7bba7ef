// pool_prepare.h -- search -> MMR without a host round trip (rlr_search_diverse): the MMR pool of a query without
// lexical candidates, built on the device by one workgroup of 1024 threads.
//
// The device twin of the host code between `search` and `mmr_diversify` (csrc/engine.cpp: search_impl's candidate
// list + rlr_engine_search_with_diversity; /root/reference/src/rag_engine.rs:531-532, :544, :734): the `fetch` best rows
// by cosine become candidates with combined = w_e * cos + w_l * 0 (two rounded products, one add), ordered (combined
// desc, NaN last, row asc) -- distinct cosines can round to one combined score, so this is NOT always the cosine order
// -- cut to the first `need`.  If such a rounding tie chain reaches the last fetched row while rows remain unfetched
// the order cannot be decided from this fetch: info[1] = 2 and the host takes the widening two-call path, exactly as
// search_impl does.
//   info[0] = pool size, info[1] = status (0 ok, 1 guard-band overflow upstream, 2 boundary tie)
// Shared by index.hip (the stand-alone launch behind sort_emit / the split pipeline) and tail.hip (the fused tail's
// finish: the workgroup that holds all re-scored candidates builds the pool in the same launch).
#pragma once

#include "common.h"
#include "kernels.h"
#include "lds_select.h"
#include "sort_emit.h"

namespace rlr {

constexpr uint32_t kPoolMax = 1024, kPoolFetchMax = kPoolMax + 8;

// LDS the body needs: keys | combined | cosine
constexpr uint32_t kPoolLdsBytes = kPoolFetchMax * 8 + kPoolFetchMax * 4 * 2;

// FROM_CANDIDATES: `packed` is the re-score's unordered candidate list (n_raw entries, capacity cap) instead of
// sort_emit's output -- ordered here, one launch and one trip through memory less per search.  More than 1024 candidates (a dense band) report status 1 and the host takes
// the two-call path, as for a guard-band overflow.  COHERENT: see sort_emit.h (candidates stored by other workgroups of
// the same launch).  Called by all 1024 threads; `lds` = kPoolLdsBytes, 16-byte aligned.
// `pre`: packed[threadIdx.x] already loaded by the caller (see sort_emit.h), or null.
template <bool FROM_CANDIDATES, bool COHERENT>
__device__ inline void pool_prepare_body(const uint64_t *packed, uint32_t n_raw, uint32_t cap, const PoolArgs &pa, char *lds,
                                         const uint64_t *pre = nullptr)
{
    uint64_t *s_key = reinterpret_cast<uint64_t *>(lds);
    float *s_c = reinterpret_cast<float *>(s_key + kPoolFetchMax);
    float *s_e = s_c + kPoolFetchMax;
    __shared__ uint32_t s_got;
    __shared__ float s_cneed, s_ctail;
    const uint32_t t = threadIdx.x;
    const uint32_t fetch = pa.fetch, need = pa.need;
    if (t == 0) {
        s_got = 0;
        s_cneed = 0.0f;
        s_ctail = 0.0f;
    }
    if constexpr (FROM_CANDIDATES) {
        // ONE rank sort, by the combined key, over all the candidates.  combined(cos) = w_e * cos + w_l * 0 never decreases
        // with the cosine, so (a) the combined score of the got-th best cosine -- the bound on every unfetched row -- is the
        // got-th largest combined score, at position got - 1 of this order, and (b) whenever the need-th position beats that
        // bound (status 0) the first `need` positions hold the same rows as the first `need` of the `fetch` best cosines
        // re-ordered by combined score: a candidate beyond the fetch scores at most the bound.  (It was a rank sort by
        // cosine, the cut to `fetch`, and a second rank sort by combined score: ~4 us of a 17 us launch.)
        const bool overflow = n_raw > cap || n_raw > 1024;
        const uint32_t got = overflow ? 0u : min(n_raw, fetch);
        float c = 0.0f, e = 0.0f;
        uint64_t mine = 0;
        if (!overflow && t < n_raw) {
            const uint64_t p = pre ? *pre : load_candidate<COHERENT>(packed + t);
            e = key_score(static_cast<uint32_t>(p >> 32));
            const float t0 = pa.w_e * e;
            const float t1 = pa.w_l * 0.0f;
            c = t0 + t1;
            mine = (static_cast<uint64_t>(score_key(c)) << 32) | (p & 0xFFFFFFFFull); // unique: the row is part of the key
            s_key[t] = mine;
        }
        __syncthreads();
        if (!overflow && t < n_raw) {
            const uint32_t rank = lds_rank_desc(s_key, n_raw, mine);
            if (rank < need) {
                pa.list[rank] = 0xFFFFFFFFu - static_cast<uint32_t>(mine & 0xFFFFFFFFull);
                pa.comb[rank] = c;
                pa.cosv[rank] = e;
                if (rank == need - 1)
                    s_cneed = c;
            }
            if (rank == got - 1)
                s_ctail = c;
        }
        const uint32_t n_pool = min(got, need);
        for (uint32_t i = n_pool + t; i < need; i += 1024) { // unused slots: a valid row, never read by the greedy kernel
            pa.list[i] = 0;
            pa.comb[i] = 0.0f;
            pa.cosv[i] = 0.0f;
        }
        __syncthreads();
        if (t == 0) {
            uint32_t status = overflow ? 1u : 0u;
            if (!overflow && got < pa.n_rows && got > 0) {
                const float c_tail = s_ctail;
                const bool ok = got >= need && (c_tail != c_tail || s_cneed > c_tail);
                if (!ok)
                    status = 2u;
            }
            pa.info[0] = status ? 0u : n_pool;
            pa.info[1] = status;
        }
        return;
    }
    const bool overflow = packed[0] == ~0ull;
    __syncthreads();
    for (uint32_t i = t; i < fetch; i += 1024) {
        const uint64_t p = overflow ? 0ull : packed[i];
        uint64_t key = 0;
        if (p != 0) { // valid entries are a prefix: (score desc, row asc), padding zeros behind
            const float e = key_score(static_cast<uint32_t>(p >> 32));
            const float t0 = pa.w_e * e;
            const float t1 = pa.w_l * 0.0f;
            const float c = t0 + t1;
            s_c[i] = c;
            s_e[i] = e;
            key = (static_cast<uint64_t>(score_key(c)) << 32) | (p & 0xFFFFFFFFull);
            atomicAdd(&s_got, 1u);
        }
        s_key[i] = key;
    }
    __syncthreads();
    const uint32_t got = s_got;
    // rank sort: keys are unique (the row is part of the key)
    for (uint32_t i = t; i < got; i += 1024) {
        const uint64_t mine = s_key[i];
        const uint32_t rank = lds_rank_desc(s_key, got, mine);
        if (rank < need) {
            pa.list[rank] = 0xFFFFFFFFu - static_cast<uint32_t>(mine & 0xFFFFFFFFull);
            pa.comb[rank] = s_c[i];
            pa.cosv[rank] = s_e[i];
            if (rank == need - 1)
                s_cneed = s_c[i];
        }
    }
    const uint32_t n_pool = min(got, need);
    for (uint32_t i = n_pool + t; i < need; i += 1024) { // unused slots: a valid row, never read by the greedy kernel
        pa.list[i] = 0;
        pa.comb[i] = 0.0f;
        pa.cosv[i] = 0.0f;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t status = overflow ? 1u : 0u;
        if (!overflow && got < pa.n_rows && got > 0) {
            // bound on every unfetched row: the combined score of the last fetched cosine
            const float t0 = pa.w_e * s_e[got - 1];
            const float t1 = pa.w_l * 0.0f;
            const float c_tail = t0 + t1;
            const bool ok = got >= need && (c_tail != c_tail || s_cneed > c_tail);
            if (!ok)
                status = 2u;
        }
        pa.info[0] = status ? 0u : n_pool;
        pa.info[1] = status;
    }
}

} // namespace rlr
