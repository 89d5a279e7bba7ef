// exact_dot.h -- the reference-order dot product (dot_product,
// /root/reference/src/rag_engine.rs:1777-1779) of an LDS-resident query with one stored
// row: strict left-to-right accumulation, product rounded before the add, no FMA (every
// translation unit of this library is built with -ffp-contract=off).  One lane = one row.
#pragma once

#include "common.h"

namespace rlr {

template <bool F16>
__device__ inline float dot_ref_row(const float4 *__restrict__ row, const float *s_q, uint32_t dim)
{
    float s = 0.0f;
    constexpr uint32_t EPU = F16 ? 8 : 4; // elements per 16-byte unit
    const uint32_t full = dim / EPU;
    uint32_t u = 0;
    // 4 units (64 B) in flight per lane
    for (; u + 4 <= full; u += 4) {
        float4 x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            x[i] = row[u + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *q = s_q + (u + i) * EPU;
            if constexpr (F16) {
                const uint32_t w[4] = {__builtin_bit_cast(uint32_t, x[i].x), __builtin_bit_cast(uint32_t, x[i].y),
                                       __builtin_bit_cast(uint32_t, x[i].z), __builtin_bit_cast(uint32_t, x[i].w)};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float p0 = h2f(static_cast<uint16_t>(w[j] & 0xFFFF)) * q[2 * j];
                    s = s + p0;
                    float p1 = h2f(static_cast<uint16_t>(w[j] >> 16)) * q[2 * j + 1];
                    s = s + p1;
                }
            } else {
                float p;
                p = x[i].x * q[0]; s = s + p;
                p = x[i].y * q[1]; s = s + p;
                p = x[i].z * q[2]; s = s + p;
                p = x[i].w * q[3]; s = s + p;
            }
        }
    }
    // remaining elements one by one (also covers dim not a multiple of the unit)
    if constexpr (F16) {
        const uint16_t *h = reinterpret_cast<const uint16_t *>(row);
        for (uint32_t e = u * EPU; e < dim; ++e) {
            float p = h2f(h[e]) * s_q[e];
            s = s + p;
        }
    } else {
        const float *f = reinterpret_cast<const float *>(row);
        for (uint32_t e = u * EPU; e < dim; ++e) {
            float p = f[e] * s_q[e];
            s = s + p;
        }
    }
    return s;
}

} // namespace rlr
