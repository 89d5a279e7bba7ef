// q8.hip -- optional 8-bit nomination copy of the rows (f32 or binary16) for single-query scans.
//
// The scan of scan.hip only NOMINATES candidates; the reference-order re-score of exact.hip makes the
// result exact.  So the scan may read any approximation of the rows whose error is BOUNDED: here one byte per
// element (k = round(x / s_r), s_r = max|x| / 127 per row, stored biased by 128) plus the row's scale -- a
// quarter of the f32 bytes.  For the band the library keeps, per index,
//     delta_max = max_r || x_r - s_r k_r ||_2      (computed exactly at build time, rounded up)
// and Cauchy-Schwarz gives | q . x_r - q . (s_r k_r) | <= ||q||_2 * delta_max for every row; the f32
// evaluation of the quantised dot adds at most arith_eps (below).  A row holding a NaN gets a NaN scale, hence a
// NaN nominated score, ordered last exactly like its reference dot; a row holding an Inf (whose reference dot may
// be +inf, -inf or NaN depending on the query) switches the index back to the f32 scan.
// Everything after the scan (select, re-score from the f32 master rows, sort) is the ordinary pipeline with
// the wider band; results are identical to the f32 scan.
#include "common.h"
#include "kernels.h"
#include "../../include/rlr_gpu.h"

#include <algorithm>
#include <cstdlib>

namespace rlr {

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ inline float wave_max_abs(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

__device__ inline float wave_add_all(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off);
    return v;
}

// one wavefront per row: scale, bytes, exact quantisation error norm
template <bool F16>
__global__ __launch_bounds__(256) void q8_build_kernel(const void *__restrict__ rows, uint32_t pitch_bytes, uint32_t dim,
                                                       uint32_t row_begin, uint32_t n_rows, uint8_t *__restrict__ q8,
                                                       float *__restrict__ scale, uint32_t *__restrict__ stats)
{
    const int lane = threadIdx.x & 63;
    const uint32_t row = row_begin + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows)
        return;
    const unsigned char *rowp = static_cast<const unsigned char *>(rows) + static_cast<size_t>(row) * pitch_bytes;
    auto elem = [&](uint32_t e) -> float {
        if constexpr (F16)
            return h2f(reinterpret_cast<const uint16_t *>(rowp)[e]);
        else
            return reinterpret_cast<const float *>(rowp)[e];
    };
    float m = 0.0f;
    bool has_nan = false, has_inf = false;
    for (uint32_t e = lane; e < dim; e += 64) {
        const float v = elem(e);
        has_nan |= v != v;
        has_inf |= __builtin_fabsf(v) > 3.4028234e38f;
        m = fmaxf(m, __builtin_fabsf(v));
    }
    has_nan = __ballot(has_nan) != 0;
    has_inf = __ballot(has_inf) != 0;
    const bool bad = has_nan || has_inf;
    m = wave_max_abs(m);
    const float s = (bad || m == 0.0f) ? 1.0f : m / 127.0f;
    float err2 = 0.0f;
    uint8_t *out = q8 + static_cast<size_t>(row) * dim;
    for (uint32_t e = lane; e < dim; e += 64) {
        const float v = bad ? 0.0f : elem(e);
        float k = __builtin_rintf(v / s);
        k = fminf(fmaxf(k, -127.0f), 127.0f);
        const float d = v - s * k;
        err2 += d * d;
        out[e] = static_cast<uint8_t>(static_cast<int>(k) + 128);
    }
    err2 = wave_add_all(err2);
    if (lane == 0) {
        // a NaN element makes the reference dot NaN (ordered last): the scan reproduces that through a NaN scale.
        // An Inf element can make it +inf, -inf or NaN depending on the query: no bounded nomination exists, the
        // index stops using the 8-bit copy (stats[2]).
        scale[row] = bad ? __builtin_bit_cast(float, 0x7FC00000u) : s;
        if (has_inf && !has_nan)
            atomicOr(&stats[2], 1u);
        // rounded up: the f32 sum of dim squares is within dim * 2^-24 relative of the exact one
        const float delta = __builtin_sqrtf(err2) * 1.001f + 1e-30f;
        if (!bad) {
            atomicMax(&stats[0], __builtin_bit_cast(uint32_t, delta)); // non-negative floats order like their bits
            atomicMax(&stats[1], __builtin_bit_cast(uint32_t, s));
        }
    }
}

// byte b of dword w as float (v_cvt_f32_ubyte{0..3})
template <int B>
__device__ inline float ubyte_f32(uint32_t w)
{
    return static_cast<float>((w >> (8 * B)) & 0xFFu);
}

__device__ inline float dot16_u8(u32x4 x, const float (&q)[16], float acc)
{
    acc = __builtin_fmaf(ubyte_f32<0>(x[0]), q[0], acc);
    acc = __builtin_fmaf(ubyte_f32<1>(x[0]), q[1], acc);
    acc = __builtin_fmaf(ubyte_f32<2>(x[0]), q[2], acc);
    acc = __builtin_fmaf(ubyte_f32<3>(x[0]), q[3], acc);
    acc = __builtin_fmaf(ubyte_f32<0>(x[1]), q[4], acc);
    acc = __builtin_fmaf(ubyte_f32<1>(x[1]), q[5], acc);
    acc = __builtin_fmaf(ubyte_f32<2>(x[1]), q[6], acc);
    acc = __builtin_fmaf(ubyte_f32<3>(x[1]), q[7], acc);
    acc = __builtin_fmaf(ubyte_f32<0>(x[2]), q[8], acc);
    acc = __builtin_fmaf(ubyte_f32<1>(x[2]), q[9], acc);
    acc = __builtin_fmaf(ubyte_f32<2>(x[2]), q[10], acc);
    acc = __builtin_fmaf(ubyte_f32<3>(x[2]), q[11], acc);
    acc = __builtin_fmaf(ubyte_f32<0>(x[3]), q[12], acc);
    acc = __builtin_fmaf(ubyte_f32<1>(x[3]), q[13], acc);
    acc = __builtin_fmaf(ubyte_f32<2>(x[3]), q[14], acc);
    acc = __builtin_fmaf(ubyte_f32<3>(x[3]), q[15], acc);
    return acc;
}

// dim <= 2048, dim % 16 == 0: a row is one (two above 1024-d) 16-byte load per lane (lanes >= dim/16 idle), the lane's query
// values live in registers.  R rows in flight per wave, groups of <= 64 rows, scores parked per lane as in scan.hip.
template <int R, int L>
__global__ __launch_bounds__(256) void q8_scan_kernel(const uint8_t *__restrict__ q8, const float *__restrict__ scale,
                                                      const float *__restrict__ query, float *__restrict__ scores,
                                                      uint32_t *__restrict__ g_hist, uint32_t n_rows, uint32_t dim,
                                                      uint32_t group_rows)
{
    __shared__ uint32_t s_hist[kHistBins];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kHistBins; i += 256)
        s_hist[i] = 0;
    const uint32_t p16 = dim / 16;
    bool active[L];
    float qv[L][16];
    float qs = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        active[l] = static_cast<uint32_t>(lane + 64 * l) < p16;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            qv[l][b] = active[l] ? query[(lane + 64 * l) * 16 + b] : 0.0f;
            qs += qv[l][b];
        }
    }
    const float bias = 128.0f * qs; // the bytes are biased by 128: subtract 128 * (this lane's query slice sums)
    __syncthreads();

    const uint32_t n_groups = (n_rows + group_rows - 1) / group_rows;
    const uint32_t n_waves = gridDim.x * 4;
    for (uint32_t g = blockIdx.x * 4 + wave; g < n_groups; g += n_waves) {
        const uint32_t row0 = g * group_rows;
        const uint32_t nr = min(group_rows, n_rows - row0);
        float mine = 0.0f;
        for (uint32_t r = 0; r < nr; r += R) {
            u32x4 x[R][L];
            float sc[R];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const uint32_t row = min(row0 + r + rr, row0 + nr - 1);
                const u32x4 *p = reinterpret_cast<const u32x4 *>(q8 + static_cast<size_t>(row) * dim) + lane;
#pragma unroll
                for (int l = 0; l < L; ++l)
                    x[rr][l] = active[l] ? __builtin_nontemporal_load(p + 64 * l)
                                         : u32x4{0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
                sc[rr] = scale[row];
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                float acc = -bias;
#pragma unroll
                for (int l = 0; l < L; ++l)
                    acc = dot16_u8(x[rr][l], qv[l], acc);
                const float tot = wave_sum(acc);
                const float v = sc[rr] * tot;
                if (static_cast<uint32_t>(lane) == r + rr)
                    mine = v; // NaN scale -> NaN score -> ordered last, like the reference dot of a row holding a NaN
            }
        }
        if (static_cast<uint32_t>(lane) < nr) {
            scores[row0 + lane] = mine;
            if (g_hist)
                atomicAdd(&s_hist[score_key(mine) >> 21], 1u);
        }
    }
    if (g_hist) {
        __syncthreads();
        for (int i = tid; i < kHistBins; i += 256) {
            const uint32_t c = s_hist[i];
            if (c)
                atomicAdd(&g_hist[i], c);
        }
    }
}

// Lane-packed variant, the default at 768 / 384 / 1536 / 512 / 256 / 128-d (RLR_Q8_PACKED=0 goes back to the kernel above; measured
// 1.263 ms against 1.365 ms at 10 M x 768 on the same box, identical candidates and results):
// when dim/16 does not divide 64 the kernel above leaves lanes idle on every row load (768-d: 48 of 64 carry
// data).  Here G consecutive rows form one block of G * dim bytes = NL full wave loads (768-d: 4 rows = 3 loads;
// 384-d: 8 rows = 3 loads; 1536-d: 2 rows = 3 loads), lane l of load j holds 16-byte unit u = 64 j + l of the
// block, i.e. row u / p16, segment u % p16 -- a quarter fewer loads, converts and FMAs at 768-d.  The G row sums
// come out of G wave reductions over the lanes' parts selected by row.  Same bound as above (q8_arith_eps): 16
// chained FMAs from the bias, at most NL - 1 extra adds, 6 reduction levels, the scale.
template <int G, int NL, int NB>
__global__ __launch_bounds__(256) void q8_scan_packed_kernel(const uint8_t *__restrict__ q8, const float *__restrict__ scale,
                                                             const float *__restrict__ query, float *__restrict__ scores,
                                                             uint32_t *__restrict__ g_hist, uint32_t n_rows, uint32_t dim,
                                                             uint32_t group_rows)
{
    constexpr int RI = G * NB; // rows per iteration (NB blocks = NB * NL loads in flight per wave)
    static_assert(RI <= 64, "a group parks one score per lane");
    __shared__ uint32_t s_hist[kHistBins];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kHistBins; i += 256)
        s_hist[i] = 0;
    const uint32_t p16 = dim / 16; // host guarantees G * p16 == NL * 64 and group_rows % RI == 0, group_rows <= 64
    uint32_t rib[NL], off[NL];
    float qv[NL][16], nbias[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const uint32_t u = static_cast<uint32_t>(j * 64 + lane);
        rib[j] = u / p16;
        const uint32_t seg = u - rib[j] * p16;
        off[j] = seg * 16;
        float qs = 0.0f;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            qv[j][b] = query[seg * 16 + b];
            qs += qv[j][b];
        }
        nbias[j] = -128.0f * qs; // the bytes are biased by 128
    }
    __syncthreads();

    const uint32_t n_groups = (n_rows + group_rows - 1) / group_rows;
    const uint32_t n_waves = gridDim.x * 4;
    for (uint32_t g = blockIdx.x * 4 + wave; g < n_groups; g += n_waves) {
        const uint32_t row0 = g * group_rows;
        const uint32_t nr = min(group_rows, n_rows - row0);
        const uint32_t last = row0 + nr - 1;
        // the row scales of the whole group in one load (lane i: row row0 + i), applied when the scores are stored
        const float my_scale = scale[min(row0 + static_cast<uint32_t>(lane), last)];
        float mine = 0.0f;
        for (uint32_t r = 0; r < nr; r += RI) {
            u32x4 x[NB][NL];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb)
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    // rows past the group end are clamped to its last row (a valid address; the lane that would
                    // receive the result is never stored)
                    const uint32_t row = min(row0 + r + bb * G + rib[j], last);
                    x[bb][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(q8 + static_cast<size_t>(row) * dim + off[j]));
                }
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                float acc[NL];
#pragma unroll
                for (int j = 0; j < NL; ++j)
                    acc[j] = dot16_u8(x[bb][j], qv[j], nbias[j]);
#pragma unroll
                for (int rb = 0; rb < G; ++rb) {
                    float part = 0.0f;
#pragma unroll
                    for (int j = 0; j < NL; ++j)
                        part += rib[j] == static_cast<uint32_t>(rb) ? acc[j] : 0.0f;
                    const float tot = wave_sum(part);
                    if (static_cast<uint32_t>(lane) == r + bb * G + rb)
                        mine = tot;
                }
            }
        }
        if (static_cast<uint32_t>(lane) < nr) {
            const float v = my_scale * mine; // NaN scale -> NaN score -> ordered last, like the reference dot of a NaN row
            scores[row0 + lane] = v;
            if (g_hist)
                atomicAdd(&s_hist[score_key(v) >> 21], 1u);
        }
    }
    if (g_hist) {
        __syncthreads();
        for (int i = tid; i < kHistBins; i += 256) {
            const uint32_t c = s_hist[i];
            if (c)
                atomicAdd(&g_hist[i], c);
        }
    }
}

} // namespace

hipError_t launch_q8_build(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, uint32_t row_begin, uint32_t n_rows,
                           void *q8, float *scale, uint32_t *stats, hipStream_t s)
{
    if (row_begin >= n_rows)
        return hipSuccess;
    const uint32_t blocks = (n_rows - row_begin + 3) / 4;
    if (dtype == RLR_F16)
        hipLaunchKernelGGL(q8_build_kernel<true>, dim3(blocks), dim3(256), 0, s, rows, pitch16 * 16, dim, row_begin, n_rows,
                           static_cast<uint8_t *>(q8), scale, stats);
    else
        hipLaunchKernelGGL(q8_build_kernel<false>, dim3(blocks), dim3(256), 0, s, rows, pitch16 * 16, dim, row_begin, n_rows,
                           static_cast<uint8_t *>(q8), scale, stats);
    return hipGetLastError();
}

hipError_t launch_q8_scan(const void *q8, const float *scale, uint32_t n_rows, uint32_t dim, const float *query,
                          float *scores, uint32_t *hist, int n_cu, hipStream_t s)
{
    if (n_rows == 0)
        return hipSuccess;
    static const int tune = [] {
        const char *v = getenv("RLR_Q8_VARIANT"); // rows in flight | workgroups per CU << 8 | group rows << 16
        return v ? static_cast<int>(strtol(v, nullptr, 0)) : 0;
    }();
    const int r = (tune & 0xFF) ? (tune & 0xFF) : 8;
    const uint32_t bpc = ((tune >> 8) & 0xFF) ? ((tune >> 8) & 0xFF) : 8;
    // 8 rows in flight x 8 workgroups per CU x 32-row groups: 6.13 TB/s at 10 M x 768; 4-16 rows, 4-16 workgroups
    // and 16/64-row groups measured 5.86-6.13 (scratch/sweep_q8.sh)
    uint32_t group = ((tune >> 16) & 0xFF) ? ((tune >> 16) & 0xFF) : 32;
    group = std::max<uint32_t>(static_cast<uint32_t>(r), group / r * r);
    const uint32_t n_groups = (n_rows + group - 1) / group;
    const uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>((n_groups + 3) / 4, static_cast<uint32_t>(n_cu) * bpc));
    const uint8_t *p = static_cast<const uint8_t *>(q8);
    static const bool packed = [] {
        const char *v = getenv("RLR_Q8_PACKED");
        return !(v && v[0] == '0');
    }();
    if (packed) {
        const uint32_t p16 = dim / 16;
        // G rows per block, NL loads per block, NB blocks per iteration, rows per group, workgroups per CU (when
        // RLR_Q8_VARIANT does not say).  768-d: 91 VGPRs, 4 / 5 / 10 workgroups per CU measured 1.263 / 1.275 / 1.276 ms.
        // The NL = 1 widths keep 8 loads in flight per wave like the kernel above, with every lane carrying data.
#define RLR_Q8_PACKED_CASE(G_, NL_, NB_, GROUP_, BPC_)                                                                \
    if (p16 * G_ == NL_ * 64) {                                                                                      \
        const uint32_t pg = GROUP_;                                                                                  \
        const uint32_t pgroups = (n_rows + pg - 1) / pg;                                                             \
        const uint32_t pbpc = ((tune >> 8) & 0xFF) ? ((tune >> 8) & 0xFF) : BPC_;                                    \
        const uint32_t pblocks = std::max<uint32_t>(1, std::min<uint32_t>((pgroups + 3) / 4, static_cast<uint32_t>(n_cu) * pbpc)); \
        hipLaunchKernelGGL((q8_scan_packed_kernel<G_, NL_, NB_>), dim3(pblocks), dim3(256), 0, s, p, scale, query, scores, \
                           hist, n_rows, dim, pg);                                                                   \
        return hipGetLastError();                                                                                    \
    }
        RLR_Q8_PACKED_CASE(4, 3, 2, 32, 2) // 768-d: 4 rows = 3 loads, 8 rows per iteration; 2 workgroups per CU 1.19 ms, 3-4: 1.22-1.26, 1: 1.73
        RLR_Q8_PACKED_CASE(8, 3, 1, 32, 4) // 384-d: 8 rows = 3 loads
        RLR_Q8_PACKED_CASE(2, 3, 4, 32, 4) // 1536-d: 2 rows = 3 loads
        RLR_Q8_PACKED_CASE(2, 1, 8, 32, 8) // 512-d: 2 rows per load, 16 rows per iteration
        RLR_Q8_PACKED_CASE(4, 1, 8, 32, 8) // 256-d: 4 rows per load, 32 rows per iteration
        RLR_Q8_PACKED_CASE(8, 1, 8, 64, 8) // 128-d: 8 rows per load, 64 rows per iteration
#undef RLR_Q8_PACKED_CASE
    }
    if (dim > 1024) { // two 16-byte loads per lane per row (dim <= 2048)
        hipLaunchKernelGGL((q8_scan_kernel<4, 2>), dim3(blocks), dim3(256), 0, s, p, scale, query, scores, hist, n_rows, dim, group);
        return hipGetLastError();
    }
    switch (r) {
    case 4: hipLaunchKernelGGL((q8_scan_kernel<4, 1>), dim3(blocks), dim3(256), 0, s, p, scale, query, scores, hist, n_rows, dim, group); break;
    case 16: hipLaunchKernelGGL((q8_scan_kernel<16, 1>), dim3(blocks), dim3(256), 0, s, p, scale, query, scores, hist, n_rows, dim, group); break;
    default: hipLaunchKernelGGL((q8_scan_kernel<8, 1>), dim3(blocks), dim3(256), 0, s, p, scale, query, scores, hist, n_rows, dim, group); break;
    }
    return hipGetLastError();
}

// | f32 evaluation of (scale * sum (byte - 128) * q) - exact value | for ||q||_2 <= q_norm: every intermediate is
// at most 256 * ||q||_1 <= 256 * sqrt(dim) * q_norm, 16 chained FMAs + 6 reduction levels + bias and scale
// roundings (< 32 roundings), times the largest row scale.
float q8_arith_eps(uint32_t dim, float scale_max, float q_norm)
{
    return 32.0f * 5.9604645e-8f * 256.0f * __builtin_sqrtf(static_cast<float>(dim)) * q_norm * scale_max * 1.0625f;
}

} // namespace rlr
