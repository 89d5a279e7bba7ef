// staged_dot.h -- the reference-order dot product of a handful of candidate rows with an LDS-resident query, with the
// rows staged through LDS (coalesced 16-byte loads, products in parallel, one strict left-to-right chain per row):
// dot_product, /root/reference/src/rag_engine.rs:1777-1779.  Shared by exact.hip (the stand-alone re-score kernels) and
// tail.hip (the fused select -> re-score -> sort tail of a single-query search).  Every translation unit that includes
// this is built with -ffp-contract=off: `s = s + p` below is a rounded add of an already rounded product.
#pragma once

#include "common.h"

namespace rlr {

// Re-score with the work split the only way the reference order allows: the PRODUCTS x_i*q_i are
// independent (each is rounded once, exactly as the reference rounds it), so all 256 threads
// compute them straight from coalesced 16-byte row loads into LDS; only the ADDS are ordered, and
// one lane per candidate then runs the strict left-to-right chain s = s + p_i out of LDS with the
// reads software-pipelined two groups ahead.  ~6 us for the ~100 candidates of a top-100 query
// (one uncoalesced lane per row took ~39 us).  The same launch clears the two radix histograms
// for the next query of this context.
__device__ inline float chain_sum_lds(const float4 *__restrict__ p4, const float *__restrict__ p, uint32_t dim)
{
    float s = 0.0f;
    const uint32_t units = dim / 4;
    uint32_t u = 0;
    float4 a[4], b[4];
    if (units >= 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[i] = p4[i];
    }
    for (; u + 8 <= units; u += 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            b[i] = p4[u + 4 + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s = s + a[i].x; s = s + a[i].y; s = s + a[i].z; s = s + a[i].w;
        }
        if (u + 12 <= units) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = p4[u + 8 + i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s = s + b[i].x; s = s + b[i].y; s = s + b[i].z; s = s + b[i].w;
        }
    }
    if (u + 4 <= units) { // `a` holds units u..u+3 (loaded by the prologue or the last iteration)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s = s + a[i].x; s = s + a[i].y; s = s + a[i].z; s = s + a[i].w;
        }
        u += 4;
    }
    for (uint32_t e = u * 4; e < dim; ++e)
        s = s + p[e];
    return s;
}

// Products of `cnt` candidate rows (row numbers in s_cand, LDS) with the query in s_q, staged in
// s_p (cnt x p_pitch floats), then the reference-order sum of candidate `tid` (valid for tid < cnt).
// Called by all NT threads of the workgroup; contains one barrier.
template <bool F16, int NT = 256>
__device__ __forceinline__ float staged_reference_dot(const float4 *__restrict__ rows, uint32_t pitch16, uint32_t dim,
                                                      const float *s_q, float *s_p, const uint32_t *s_cand, uint32_t cnt,
                                                      uint32_t tid)
{
    const uint32_t q_floats = (dim + 7) & ~7u;
    const uint32_t p_pitch = q_floats + 4; // product row pitch in floats (+16 B: bank spread)
    // products: 16-byte units of the candidate rows, coalesced, kBatch independent loads in flight
    // per thread before the first is consumed; pad columns multiply to 0 and are never summed
    // (the chain stops at dim)
    const uint32_t units = F16 ? q_floats / 8 : q_floats / 4; // units that hold real columns
    const uint32_t total = cnt * units;
    constexpr int kBatch = 6;
    for (uint32_t idx0 = tid; idx0 < total; idx0 += NT * kBatch) {
        float4 x[kBatch];
        uint32_t ci[kBatch], uu[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const uint32_t idx = idx0 + NT * j;
            ci[j] = idx / units;
            uu[j] = idx - ci[j] * units;
            if (idx < total)
                x[j] = rows[static_cast<size_t>(s_cand[ci[j]]) * pitch16 + uu[j]];
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            if (idx0 + NT * j >= total)
                continue;
            const uint32_t u = uu[j];
            float *dst = s_p + ci[j] * p_pitch;
            if constexpr (F16) {
                const uint32_t w[4] = {__builtin_bit_cast(uint32_t, x[j].x), __builtin_bit_cast(uint32_t, x[j].y),
                                       __builtin_bit_cast(uint32_t, x[j].z), __builtin_bit_cast(uint32_t, x[j].w)};
                const float *q = s_q + u * 8;
                float4 lo, hi;
                lo.x = h2f(static_cast<uint16_t>(w[0] & 0xFFFF)) * q[0];
                lo.y = h2f(static_cast<uint16_t>(w[0] >> 16)) * q[1];
                lo.z = h2f(static_cast<uint16_t>(w[1] & 0xFFFF)) * q[2];
                lo.w = h2f(static_cast<uint16_t>(w[1] >> 16)) * q[3];
                hi.x = h2f(static_cast<uint16_t>(w[2] & 0xFFFF)) * q[4];
                hi.y = h2f(static_cast<uint16_t>(w[2] >> 16)) * q[5];
                hi.z = h2f(static_cast<uint16_t>(w[3] & 0xFFFF)) * q[6];
                hi.w = h2f(static_cast<uint16_t>(w[3] >> 16)) * q[7];
                reinterpret_cast<float4 *>(dst)[2 * u] = lo;
                reinterpret_cast<float4 *>(dst)[2 * u + 1] = hi;
            } else {
                const float4 q = reinterpret_cast<const float4 *>(s_q)[u];
                float4 pr;
                pr.x = x[j].x * q.x;
                pr.y = x[j].y * q.y;
                pr.z = x[j].z * q.z;
                pr.w = x[j].w * q.w;
                reinterpret_cast<float4 *>(dst)[u] = pr;
            }
        }
    }
    __syncthreads();
    float sc = 0.0f;
    if (tid < cnt) {
        const float *pr = s_p + tid * p_pitch;
        sc = chain_sum_lds(reinterpret_cast<const float4 *>(pr), pr, dim);
    }
    return sc;
}

} // namespace rlr
