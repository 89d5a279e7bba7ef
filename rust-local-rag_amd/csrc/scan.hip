// scan.hip -- the HBM-bound hot loop: one pass over the chunk-embedding matrix,
// one wavefront-order dot product per row (replaces HOT LOOP 1,
// /root/reference/src/rag_engine.rs:524-541, `dot_product` :1777-1779).
//
// Shape of the work on gfx950
//   * query staged in LDS once per workgroup, then held in VGPRs (3 x float4 per lane
//     for 768-d) for the whole kernel;
//   * a wave owns a group of <= 64 consecutive rows; each row is read with non-temporal
//     `global_load_dwordx4 ... nt`, 64 lanes x 16 B = 1 KiB per instruction, fully coalesced,
//     R rows (R x dim x 4 B) in flight per wave;
//   * per-lane fmaf partials, DPP wavefront reduction (no LDS), the row's score is
//     parked in lane (row % group) so the group's scores leave as one coalesced store;
//   * the first radix-select histogram (top 11 key bits) is accumulated in LDS while
//     the rows stream and flushed with one atomic per touched bin per workgroup.
// Algorithmic traffic: dim x elem bytes per row read once, 4 B per row written.
// There is no inter-workgroup reuse, so no XCD-aware remap is needed here (the query is
// the only shared operand: 3 KB, resident in every XCD's L2).
//
// The summation order differs from the reference's strict left-to-right order, so these
// scores only NOMINATE candidates (guard band, select.hip); exact.hip re-scores them.
#include <algorithm>
#include <cstring>

#include "common.h"
#include "kernels.h"
#include "../../include/rlr_gpu.h"

namespace rlr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ inline float4 ld16(const float4 *p)
{
    if constexpr (NT) {
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}

__device__ inline float dot4(float4 x, float4 q, float acc)
{
    acc = __builtin_fmaf(x.x, q.x, acc);
    acc = __builtin_fmaf(x.y, q.y, acc);
    acc = __builtin_fmaf(x.z, q.z, acc);
    acc = __builtin_fmaf(x.w, q.w, acc);
    return acc;
}

// 8 binary16 values (one 16-byte load) against 8 f32 query values.
__device__ inline float dot8h(float4 raw, float4 q0, float4 q1, float acc)
{
    uint32_t w0 = __builtin_bit_cast(uint32_t, raw.x), w1 = __builtin_bit_cast(uint32_t, raw.y);
    uint32_t w2 = __builtin_bit_cast(uint32_t, raw.z), w3 = __builtin_bit_cast(uint32_t, raw.w);
    acc = __builtin_fmaf(h2f(static_cast<uint16_t>(w0 & 0xFFFF)), q0.x, acc);
    acc = __builtin_fmaf(h2f(static_cast<uint16_t>(w0 >> 16)), q0.y, acc);
    acc = __builtin_fmaf(h2f(static_cast<uint16_t>(w1 & 0xFFFF)), q0.z, acc);
    acc = __builtin_fmaf(h2f(static_cast<uint16_t>(w1 >> 16)), q0.w, acc);
    acc = __builtin_fmaf(h2f(static_cast<uint16_t>(w2 & 0xFFFF)), q1.x, acc);
    acc = __builtin_fmaf(h2f(static_cast<uint16_t>(w2 >> 16)), q1.y, acc);
    acc = __builtin_fmaf(h2f(static_cast<uint16_t>(w3 & 0xFFFF)), q1.z, acc);
    acc = __builtin_fmaf(h2f(static_cast<uint16_t>(w3 >> 16)), q1.w, acc);
    return acc;
}

__device__ inline void hist_flush(const uint32_t *s_hist, uint32_t *g_hist)
{
    for (int i = threadIdx.x; i < kHistBins; i += blockDim.x) {
        uint32_t c = s_hist[i];
        if (c)
            atomicAdd(&g_hist[i], c);
    }
}

// ---------------------------------------------------------------------------
// Fixed-shape kernel: pitch16 == CH * 64 (f32: dim = 256*CH; f16: dim = 512*CH), so a
// row is exactly CH wave-wide 16-byte loads and the query lives in registers.
// ---------------------------------------------------------------------------
// The query as a kernel ARGUMENT (f32 rows, CH <= 3: 3 KB of the 4 KB a launch may carry): see ScanArgs::query_host.
template <int CH>
struct QueryArg {
    float4 v[CH * 64];
};

template <int CH, int R, bool F16, bool NT, bool KQ = false>
__global__ __launch_bounds__(256) void scan_fixed_kernel(const float4 *__restrict__ rows,
                                                         const float *__restrict__ query,
                                                         float *__restrict__ scores,
                                                         uint32_t *__restrict__ g_hist,
                                                         uint32_t n_rows, uint32_t group_rows,
                                                         QueryArg<KQ ? CH : 0> qarg = QueryArg<KQ ? CH : 0>())
{
    constexpr int P16 = CH * 64;                  // 16-byte units per row
    constexpr int QF4 = F16 ? 2 * P16 : P16;      // float4 units of query
    __shared__ float4 s_q[QF4];
    __shared__ uint32_t s_hist[kHistBins];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform -> scalar row addressing
    if constexpr (KQ) {
        // from the argument segment; workgroup 0 also leaves it in device memory for the kernels behind the scan
        for (int i = tid; i < QF4; i += 256) {
            const float4 v = qarg.v[i];
            s_q[i] = v;
            if (blockIdx.x == 0)
                reinterpret_cast<float4 *>(const_cast<float *>(query))[i] = v;
        }
    } else {
        for (int i = tid; i < QF4; i += 256)
            s_q[i] = reinterpret_cast<const float4 *>(query)[i];
    }
    for (int i = tid; i < kHistBins; i += 256)
        s_hist[i] = 0;
    __syncthreads();

    float4 qv[F16 ? 2 * CH : CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if constexpr (F16) {
            qv[2 * c] = s_q[2 * (c * 64 + lane)];
            qv[2 * c + 1] = s_q[2 * (c * 64 + lane) + 1];
        } else {
            qv[c] = s_q[c * 64 + lane];
        }
    }

    const uint32_t n_groups = (n_rows + group_rows - 1) / group_rows;
    const uint32_t n_waves = gridDim.x * 4;
    for (uint32_t g = blockIdx.x * 4 + wave; g < n_groups; g += n_waves) {
        const uint32_t row0 = g * group_rows;
        const uint32_t nr = min(group_rows, n_rows - row0);
        float mine = 0.0f;
        for (uint32_t r = 0; r < nr; r += R) {
            float4 x[R][CH];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                // rows past the group end are clamped to its last row: a valid address
                // whose result lands in a lane that is never stored.
                const uint32_t row = min(row0 + r + rr, row0 + nr - 1);
                const float4 *p = rows + static_cast<size_t>(row) * P16 + lane;
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    x[rr][c] = ld16<NT>(p + c * 64);
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                float acc = 0.0f;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if constexpr (F16)
                        acc = dot8h(x[rr][c], qv[2 * c], qv[2 * c + 1], acc);
                    else
                        acc = dot4(x[rr][c], qv[c], acc);
                }
                const float tot = wave_sum(acc);
                if (static_cast<uint32_t>(lane) == r + rr)
                    mine = tot;
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
        hist_flush(s_hist, g_hist);
    }
}

// ---------------------------------------------------------------------------
// Multi-query kernel: 2..8 queries share ONE pass over the rows.  Same row stream as the fixed kernel (f32 rows,
// pitch a multiple of 1 KiB); the Q queries sit in registers (Q x CH float4 per lane), every loaded row feeds Q
// FMA chains and Q DPP reductions -- at 768-d eight queries cost ~2.7 ms of VALU time, still under the 4.6 ms
// the HBM stream takes, so a small concurrent batch costs about one scan.  Scores go to Q arrays (stride
// score_stride); the radix histograms are left to the batched select that follows.
// ---------------------------------------------------------------------------
template <int CH, int Q>
__global__ __launch_bounds__(256) void scan_multi_kernel(const float4 *__restrict__ rows, const float *__restrict__ queries,
                                                         uint32_t q_pitch, uint32_t n_queries, float *__restrict__ scores,
                                                         size_t score_stride, uint32_t n_rows, uint32_t group_rows)
{
    constexpr int P16 = CH * 64;
    constexpr int R = 2;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float4 qv[Q][CH];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int c = 0; c < CH; ++c)
            qv[q][c] = static_cast<uint32_t>(q) < n_queries
                           ? reinterpret_cast<const float4 *>(queries + static_cast<size_t>(q) * q_pitch)[c * 64 + lane]
                           : make_float4(0.0f, 0.0f, 0.0f, 0.0f);

    const uint32_t n_groups = (n_rows + group_rows - 1) / group_rows;
    const uint32_t n_waves = gridDim.x * 4;
    for (uint32_t g = blockIdx.x * 4 + wave; g < n_groups; g += n_waves) {
        const uint32_t row0 = g * group_rows;
        const uint32_t nr = min(group_rows, n_rows - row0);
        float mine[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q)
            mine[q] = 0.0f;
        for (uint32_t r = 0; r < nr; r += R) {
            float4 x[R][CH];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const uint32_t row = min(row0 + r + rr, row0 + nr - 1);
                const float4 *p = rows + static_cast<size_t>(row) * P16 + lane;
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    x[rr][c] = ld16<true>(p + c * 64);
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    float acc = 0.0f;
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        acc = dot4(x[rr][c], qv[q][c], acc);
                    const float tot = wave_sum(acc);
                    if (static_cast<uint32_t>(lane) == r + rr)
                        mine[q] = tot;
                }
            }
        }
        if (static_cast<uint32_t>(lane) < nr) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (static_cast<uint32_t>(q) < n_queries)
                    scores[static_cast<size_t>(q) * score_stride + row0 + lane] = mine[q];
        }
    }
}

// ---------------------------------------------------------------------------
// Packed kernel: row pitch a multiple of 256 B but not of 1 KiB (384-d f32, 768-d f16, ...).
// G = 64 / gcd(P16, 64) consecutive rows form a contiguous "pack" of exactly M = G * P16 / 64
// wave-wide 16-byte loads, so every load instruction is still 64 lanes x 16 B of consecutive
// addresses; a lane's position inside the pack decides which row and which query column it
// serves (fixed per lane: the query stays in registers), loads that straddle two rows feed two
// accumulators through a lane mask, and each row still costs one DPP wave reduction.
// ---------------------------------------------------------------------------
constexpr int pack_gcd(int a, int b) { return b == 0 ? a : pack_gcd(b, a % b); }

template <int P16>
struct PackShape {
    static constexpr int G = 64 / pack_gcd(P16, 64);        // rows per pack
    static constexpr int M = G * P16 / 64;                   // loads per pack
    static constexpr int R = M >= 9 ? 1 : (M >= 5 ? 2 : (M >= 2 ? 4 : 8)); // packs in flight (8-14 loads per lane)
};

template <int P16, bool F16>
__global__ __launch_bounds__(256) void scan_packed_kernel(const float4 *__restrict__ rows,
                                                          const float *__restrict__ query,
                                                          float *__restrict__ scores,
                                                          uint32_t *__restrict__ g_hist,
                                                          uint32_t n_rows, uint32_t group_rows)
{
    constexpr int G = PackShape<P16>::G, M = PackShape<P16>::M, R = PackShape<P16>::R;
    constexpr int QF4 = F16 ? 2 * P16 : P16;
    __shared__ float4 s_q[QF4];
    __shared__ uint32_t s_hist[kHistBins];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < QF4; i += 256)
        s_q[i] = reinterpret_cast<const float4 *>(query)[i];
    for (int i = tid; i < kHistBins; i += 256)
        s_hist[i] = 0;
    __syncthreads();

    float4 qv[F16 ? 2 * M : M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const int col = (j * 64 + lane) % P16;
        if constexpr (F16) {
            qv[2 * j] = s_q[2 * col];
            qv[2 * j + 1] = s_q[2 * col + 1];
        } else {
            qv[j] = s_q[col];
        }
    }

    const size_t last_unit = static_cast<size_t>(n_rows) * P16 - 1; // loads past the matrix end re-read its last 16 B
    const uint32_t n_groups = (n_rows + group_rows - 1) / group_rows;
    const uint32_t n_waves = gridDim.x * 4;
    for (uint32_t g = blockIdx.x * 4 + wave; g < n_groups; g += n_waves) {
        const uint32_t row0 = g * group_rows;
        const uint32_t nr = min(group_rows, n_rows - row0);
        float mine = 0.0f;
        for (uint32_t r = 0; r < nr; r += R * G) {
            float4 x[R][M];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                // packs past the group end are clamped to its last pack (their lanes are never stored)
                const uint32_t pack_row = min(row0 + r + rr * G, row0 + ((nr - 1) / G) * G);
                const size_t base = static_cast<size_t>(pack_row) * P16 + lane;
#pragma unroll
                for (int j = 0; j < M; ++j)
                    x[rr][j] = ld16<true>(rows + min(base + j * 64, last_unit));
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                float acc[G];
#pragma unroll
                for (int k = 0; k < G; ++k)
                    acc[k] = 0.0f;
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    const int g_lo = (j * 64) / P16, g_hi = (j * 64 + 63) / P16;
                    if (g_lo == g_hi) {
                        if constexpr (F16)
                            acc[g_lo] = dot8h(x[rr][j], qv[2 * j], qv[2 * j + 1], acc[g_lo]);
                        else
                            acc[g_lo] = dot4(x[rr][j], qv[j], acc[g_lo]);
                    } else {
                        float t;
                        if constexpr (F16)
                            t = dot8h(x[rr][j], qv[2 * j], qv[2 * j + 1], 0.0f);
                        else
                            t = dot4(x[rr][j], qv[j], 0.0f);
                        const int lane_row = (j * 64 + lane) / P16;
#pragma unroll
                        for (int k = 0; k < G; ++k)
                            if (k >= g_lo && k <= g_hi)
                                acc[k] += lane_row == k ? t : 0.0f;
                    }
                }
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const float tot = wave_sum(acc[k]);
                    if (static_cast<uint32_t>(lane) == r + rr * G + k)
                        mine = tot;
                }
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
        hist_flush(s_hist, g_hist);
    }
}

// ---------------------------------------------------------------------------
// Generic kernel: any dim (row pitch padded to 16 B, pad = 0), query read from LDS.
// ---------------------------------------------------------------------------
template <int R, bool F16>
__global__ __launch_bounds__(256) void scan_generic_kernel(const float4 *__restrict__ rows,
                                                           const float *__restrict__ query,
                                                           float *__restrict__ scores,
                                                           uint32_t *__restrict__ g_hist,
                                                           uint32_t n_rows, uint32_t group_rows,
                                                           uint32_t pitch16)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_raw);                       // kHistBins
    float4 *s_q = reinterpret_cast<float4 *>(s_raw + kHistBins * sizeof(uint32_t)); // query

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform -> scalar row addressing
    const uint32_t qf4 = F16 ? 2 * pitch16 : pitch16;
    for (uint32_t i = tid; i < qf4; i += 256)
        s_q[i] = reinterpret_cast<const float4 *>(query)[i];
    for (int i = tid; i < kHistBins; i += 256)
        s_hist[i] = 0;
    __syncthreads();

    const uint32_t n_groups = (n_rows + group_rows - 1) / group_rows;
    const uint32_t n_waves = gridDim.x * 4;
    for (uint32_t g = blockIdx.x * 4 + wave; g < n_groups; g += n_waves) {
        const uint32_t row0 = g * group_rows;
        const uint32_t nr = min(group_rows, n_rows - row0);
        float mine = 0.0f;
        for (uint32_t r = 0; r < nr; r += R) {
            float acc[R];
            const float4 *p[R];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                acc[rr] = 0.0f;
                const uint32_t row = min(row0 + r + rr, row0 + nr - 1);
                p[rr] = rows + static_cast<size_t>(row) * pitch16;
            }
            for (uint32_t c = lane; c < pitch16; c += 64) {
                float4 x[R];
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
                    x[rr] = ld16<true>(p[rr] + c);
                if constexpr (F16) {
                    const float4 q0 = s_q[2 * c], q1 = s_q[2 * c + 1];
#pragma unroll
                    for (int rr = 0; rr < R; ++rr)
                        acc[rr] = dot8h(x[rr], q0, q1, acc[rr]);
                } else {
                    const float4 q0 = s_q[c];
#pragma unroll
                    for (int rr = 0; rr < R; ++rr)
                        acc[rr] = dot4(x[rr], q0, acc[rr]);
                }
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const float tot = wave_sum(acc[rr]);
                if (static_cast<uint32_t>(lane) == r + rr)
                    mine = tot;
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
        hist_flush(s_hist, g_hist);
    }
}

struct ScanPlan {
    uint32_t group_rows;
    uint32_t blocks;
    int r;
    bool nt;
};

ScanPlan plan_scan(const ScanArgs &a)
{
    // Defaults from sweeps on MI355X (scratch/sweep_r8.sh, sweep_r8_dims.sh; GB/s at 24-30 GB corpora,
    // old default R=4 x 8 workgroups/CU -> R=8 x 4 workgroups/CU): 768-d f32 6553 -> 6702 (6797 with
    // 16-row groups), 512-d 6524 -> 6691, 1024-d 6541 -> 6837, 256-d 6437 -> 6496, f16 512/768/1024-d
    // 6327/6526/6481 -> 6563/6628/6535; the gain holds from 300 k to 10 M rows.
    ScanPlan p;
    const int v = a.variant;
    const int r_code = v & 0xF;
    // small corpora (fewer than ~3 sixteen-row groups per resident wave, < 200 k rows): more, lighter waves
    // balance better -- 100 k rows: 55 us with R=4 x 8 workgroups/CU, 63 us with R=8 x 4
    const bool small = (a.n_rows + 15) / 16 < static_cast<uint32_t>(a.n_cu) * 4 * 4 * 3;
    p.r = r_code == 0 ? (small ? 4 : 8) : r_code;
    p.nt = ((v >> 4) & 1) == 0; // non-temporal row loads by default: +10 % on MI355X (6.8 vs 6.15 TB/s); bit 4 turns them off
    int blocks_per_cu = (v >> 8) & 0xFF;
    if (blocks_per_cu == 0)
        blocks_per_cu = small ? 8 : 4;
    uint32_t group = (v >> 16) & 0xFF;
    const uint32_t max_blocks = static_cast<uint32_t>(a.n_cu) * blocks_per_cu;
    if (group == 0) {
        // enough groups to give every resident wave several, but never below 16 rows
        group = 64;
        while (group > 16 && (a.n_rows + group - 1) / group < max_blocks * 4 * 4)
            group >>= 1;
        if (a.dtype == RLR_F32 && a.pitch16 == 192)
            group = 16; // 3 KiB rows: 16-row groups measured +1.5-3 % at every corpus size
    }
    if (group > 64)
        group = 64;
    group = (group / p.r) * p.r;
    if (group == 0)
        group = p.r;
    // One workgroup per CU -- one wave per SIMD -- for the row shapes where a wave's own loads in flight (R rows x 3-6 KB) cover
    // the memory latency: fewer, longer streams reach 0.875 of the HBM peak on 10 M x 768 f32 where four workgroups per CU of
    // 8-row steps reach 0.846 (scratch/sweep_scan_10m.sh, four interleaved repeats: 4.34-4.41 against 4.53-4.56 ms), and the
    // gain grows as the corpus shrinks (1 M rows 434 against 468 us, 200 k rows 91 against 120, 100 k rows 51 against 58).
    // 1536-d rows (two-row steps): f32 7.1 against 6.6 TB/s, binary16 6.85 against 6.6.  Every other shape measured
    // (256 / 512 / 1024 / 2048-d, the packed and the generic kernel) is level or loses with one wave per SIMD
    // (scratch/sweep_scan_shapes.sh) and keeps the settings above.  Only when RLR_SCAN_VARIANT leaves all three fields open.
    // [Tried on top: the wave's row stream software-pipelined (the next step's loads issued before the current step is
    // reduced, 4..8 rows in flight all the time) -- 4.61 ms whatever the step, group or workgroup count, against 4.45 on the
    // same box: the bursts with pauses in between suit the memory system better than a steady deeper queue.  Not kept.
    // Steps of 3 / 5 / 6 rows at one workgroup per CU: 0.825 / 0.73 / 0.46 of the peak against 0.875 with 4 -- twelve 16-byte
    // loads per lane in flight is the most a lone wave gets through without stalling.]
    if (r_code == 0 && ((v >> 8) & 0xFF) == 0 && ((v >> 16) & 0xFF) == 0) {
        const uint32_t elems = a.dtype == RLR_F16 ? a.pitch16 * 8 : a.pitch16 * 4;
        if (elems == a.dim && a.dtype == RLR_F32 && a.pitch16 == 192) {
            p.r = 4;
            blocks_per_cu = 1;
            group = 8;
        } else if (elems == a.dim && a.dtype == RLR_F32 && a.pitch16 == 384) {
            blocks_per_cu = 1;
            group = 8;
        } else if (elems == a.dim && a.dtype == RLR_F16 && a.pitch16 == 192) {
            blocks_per_cu = 1;
            group = 16;
        }
    }
    p.group_rows = group;
    const uint32_t n_groups = (a.n_rows + group - 1) / group;
    uint32_t blocks = (n_groups + 3) / 4;
    if (blocks > static_cast<uint32_t>(a.n_cu) * blocks_per_cu)
        blocks = static_cast<uint32_t>(a.n_cu) * blocks_per_cu;
    if (blocks == 0)
        blocks = 1;
    p.blocks = blocks;
    return p;
}

bool kq_enabled()
{
    static const bool on = [] {
        const char *v = getenv("RLR_SCAN_KQ"); // the query in the kernel arguments (ScanArgs::query_host); "0" turns it off
        return !(v && v[0] == '0');
    }();
    return on;
}

template <int CH>
hipError_t launch_fixed_kq(const ScanArgs &a, const ScanPlan &p, hipStream_t s)
{
    const float4 *rows = static_cast<const float4 *>(a.rows);
    QueryArg<CH> q;
    std::memcpy(q.v, a.query_host, sizeof(q.v));
#define RLR_SCAN_KQ_CASE(RV)                                                                                  \
    hipLaunchKernelGGL((scan_fixed_kernel<CH, RV, false, true, true>), dim3(p.blocks), dim3(256), 0, s, rows, \
                       a.query, a.scores, a.hist, a.n_rows, p.group_rows, q)
    switch (p.r) {
    case 1: RLR_SCAN_KQ_CASE(1); break;
    case 2: RLR_SCAN_KQ_CASE(2); break;
    case 8: RLR_SCAN_KQ_CASE(8); break;
    default: RLR_SCAN_KQ_CASE(4); break;
    }
#undef RLR_SCAN_KQ_CASE
    return hipGetLastError();
}

template <int CH, bool F16>
hipError_t launch_fixed(const ScanArgs &a, const ScanPlan &p, hipStream_t s)
{
    if constexpr (!F16 && CH <= 3) {
        if (a.query_host && p.nt && kq_enabled())
            return launch_fixed_kq<CH>(a, p, s);
    }
    const float4 *rows = static_cast<const float4 *>(a.rows);
#define RLR_SCAN_CASE(RV, NTV)                                                                    \
    hipLaunchKernelGGL((scan_fixed_kernel<CH, RV, F16, NTV>), dim3(p.blocks), dim3(256), 0, s,   \
                       rows, a.query, a.scores, a.hist, a.n_rows, p.group_rows)
    if (p.nt) {
        switch (p.r) {
        case 1: RLR_SCAN_CASE(1, true); break;
        case 2: RLR_SCAN_CASE(2, true); break;
        case 8: RLR_SCAN_CASE(8, true); break;
        default: RLR_SCAN_CASE(4, true); break;
        }
    } else {
        switch (p.r) {
        case 1: RLR_SCAN_CASE(1, false); break;
        case 2: RLR_SCAN_CASE(2, false); break;
        case 8: RLR_SCAN_CASE(8, false); break;
        default: RLR_SCAN_CASE(4, false); break;
        }
    }
#undef RLR_SCAN_CASE
    return hipGetLastError();
}

bool fixed_shape(const ScanArgs &a, int *ch)
{
    if (a.pitch16 % 64 != 0)
        return false;
    const int c = a.pitch16 / 64;
    const uint32_t elems = a.dtype == RLR_F16 ? a.pitch16 * 8 : a.pitch16 * 4;
    if (elems != a.dim)
        return false;
    if (a.dtype == RLR_F16) {
        if (c < 1 || c > 4)
            return false;
    } else if (c < 1 || c > 8 || c == 7) {
        return false;
    }
    *ch = c;
    return true;
}

// wide rows (CH >= 5: 1280/1536/2048-d f32): two rows in flight keep the kernel at 8 waves/SIMD
template <int CH, bool F16>
hipError_t launch_fixed_wide(const ScanArgs &a, ScanPlan p, hipStream_t s)
{
    p.r = 2;
    p.group_rows = std::max<uint32_t>((p.group_rows / 2) * 2, 2);
    p.blocks = std::max<uint32_t>(1, std::min<uint32_t>(p.blocks, ((a.n_rows + p.group_rows - 1) / p.group_rows + 3) / 4));
    hipLaunchKernelGGL((scan_fixed_kernel<CH, 2, F16, true>), dim3(p.blocks), dim3(256), 0, s,
                       static_cast<const float4 *>(a.rows), a.query, a.scores, a.hist, a.n_rows, p.group_rows);
    return hipGetLastError();
}

template <int P16, bool F16>
hipError_t launch_packed_one(const ScanArgs &a, ScanPlan p, hipStream_t s)
{
    constexpr uint32_t step = PackShape<P16>::R * PackShape<P16>::G; // rows per loop iteration
    uint32_t group = std::max<uint32_t>(p.group_rows, step);
    group = std::min<uint32_t>(64, (group / step) * step);
    const uint32_t n_groups = (a.n_rows + group - 1) / group;
    const uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>((n_groups + 3) / 4, static_cast<uint32_t>(a.n_cu) * 8));
    hipLaunchKernelGGL((scan_packed_kernel<P16, F16>), dim3(blocks), dim3(256), 0, s, static_cast<const float4 *>(a.rows),
                       a.query, a.scores, a.hist, a.n_rows, group);
    return hipGetLastError();
}

// Row pitches served by the packed kernel: the short rows (256 B and 512 B), where the generic kernel
// leaves most lanes of a wave idle.  Measured on MI355X, 24 GB corpora, GB/s packed vs generic (both with
// non-temporal loads; scratch/packed_ab.sh):
//   64-d f32 6119/4053   128-d f32 6130/5898   128-d f16 5923/3520   256-d f16 6296/5935
// From 320-d f32 / 768-d f16 upwards the two kernels are within noise of each other (6.5-6.6 TB/s), so
// those pitches stay on the generic kernel (one instantiation instead of one per pitch).
bool launch_packed(const ScanArgs &a, const ScanPlan &p, hipStream_t s, hipError_t *e)
{
    const uint32_t elems = a.dtype == RLR_F16 ? a.pitch16 * 8 : a.pitch16 * 4;
    if (elems != a.dim || (a.variant & 0x20)) // bit 5 of RLR_SCAN_VARIANT: force the generic kernel (A/B)
        return false;
    const bool h = a.dtype == RLR_F16;
    switch (a.pitch16) {
    case 16: *e = h ? launch_packed_one<16, true>(a, p, s) : launch_packed_one<16, false>(a, p, s); return true;
    case 32: *e = h ? launch_packed_one<32, true>(a, p, s) : launch_packed_one<32, false>(a, p, s); return true;
    default: return false;
    }
}

} // namespace

bool launch_scan_takes_host_query(const ScanArgs &a)
{
    int ch = 0;
    if (!a.query_host || a.n_rows == 0 || a.dtype != RLR_F32 || !fixed_shape(a, &ch) || ch > 3 || !kq_enabled())
        return false;
    return plan_scan(a).nt;
}

hipError_t launch_scan(const ScanArgs &a, hipStream_t s)
{
    if (a.n_rows == 0)
        return hipSuccess;
    ScanPlan p = plan_scan(a);
    int ch = 0;
    if (fixed_shape(a, &ch)) {
        if (a.dtype == RLR_F16) {
            if (p.r == 8) {
                p.r = 4;
                p.group_rows = (p.group_rows / 4) * 4;
            }
            switch (ch) {
            case 1: return launch_fixed<1, true>(a, p, s);
            case 2: return launch_fixed<2, true>(a, p, s);
            case 3: return launch_fixed_wide<3, true>(a, p, s);
            default: return launch_fixed_wide<4, true>(a, p, s);
            }
        }
        switch (ch) {
        case 1: return launch_fixed<1, false>(a, p, s);
        case 2: return launch_fixed<2, false>(a, p, s);
        case 3: return launch_fixed<3, false>(a, p, s);
        case 4: return launch_fixed<4, false>(a, p, s);
        case 5: return launch_fixed_wide<5, false>(a, p, s);
        case 6: return launch_fixed_wide<6, false>(a, p, s);
        default: return launch_fixed_wide<8, false>(a, p, s);
        }
    }
    hipError_t pe = hipSuccess;
    if (launch_packed(a, p, s, &pe))
        return pe;
    // generic path
    const uint32_t qf4 = a.dtype == RLR_F16 ? 2 * a.pitch16 : a.pitch16;
    const size_t lds = kHistBins * sizeof(uint32_t) + static_cast<size_t>(qf4) * 16;
    const float4 *rows = static_cast<const float4 *>(a.rows);
    p.r = 2;
    p.group_rows = (p.group_rows / 2) * 2;
    if (p.group_rows == 0)
        p.group_rows = 2;
    if (a.dtype == RLR_F16)
        hipLaunchKernelGGL((scan_generic_kernel<2, true>), dim3(p.blocks), dim3(256), lds, s, rows,
                           a.query, a.scores, a.hist, a.n_rows, p.group_rows, a.pitch16);
    else
        hipLaunchKernelGGL((scan_generic_kernel<2, false>), dim3(p.blocks), dim3(256), lds, s, rows,
                           a.query, a.scores, a.hist, a.n_rows, p.group_rows, a.pitch16);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Read-only streaming probe (the measured-peak denominator of bench.py's roofline, SURVEY.md 8(d) "re-measure on the
// box"): the scan's row stream with the arithmetic removed -- every wave reads bursts of U consecutive KiB with
// non-temporal 16-byte loads, adds the words up and keeps the sum (one store per wave, so the loads cannot be dropped).
// ---------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(256) void probe_read_kernel(const float4 *__restrict__ p, size_t n_kib, float *__restrict__ sink)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = static_cast<size_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const size_t n_waves = static_cast<size_t>(gridDim.x) * 4;
    float acc = 0.0f;
    for (size_t b = wave * U; b < n_kib; b += n_waves * U) {
        float4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            x[u] = ld16<true>(p + min(b + u, n_kib - 1) * 64 + lane);
#pragma unroll
        for (int u = 0; u < U; ++u)
            acc += (x[u].x + x[u].y) + (x[u].z + x[u].w);
    }
    sink[wave * 64 + lane] = acc;
}

// bytes / 1024 KiB are read; sink holds blocks * 256 floats.  shape: 0 = one workgroup per CU, 12 KiB bursts (the
// headline scan's shape), 1 = four per CU, 8 KiB bursts, 2 = eight per CU, 4 KiB bursts.
hipError_t launch_probe_read(const void *p, size_t bytes, float *sink, int n_cu, int shape, hipStream_t s)
{
    const size_t n_kib = bytes / 1024;
    if (n_kib == 0)
        return hipSuccess;
    const float4 *p4 = static_cast<const float4 *>(p);
    if (shape == 0)
        hipLaunchKernelGGL(probe_read_kernel<12>, dim3(n_cu), dim3(256), 0, s, p4, n_kib, sink);
    else if (shape == 1)
        hipLaunchKernelGGL(probe_read_kernel<8>, dim3(n_cu * 4), dim3(256), 0, s, p4, n_kib, sink);
    else
        hipLaunchKernelGGL(probe_read_kernel<4>, dim3(n_cu * 8), dim3(256), 0, s, p4, n_kib, sink);
    return hipGetLastError();
}

// 2..8 queries over f32 rows whose pitch is a multiple of 1 KiB (256/512/768/1024-d); false otherwise
bool launch_scan_multi(const ScanArgs &a, uint32_t q_pitch, uint32_t n_queries, size_t score_stride, hipStream_t s,
                       hipError_t *err)
{
    int ch = 0;
    if (a.dtype != RLR_F32 || n_queries < 2 || n_queries > 8 || !fixed_shape(a, &ch) || ch > 4)
        return false;
    static const int tune = [] {
        const char *v = getenv("RLR_SCAN_MULTI_VARIANT"); // workgroups per CU | rows per group << 8 (A/B)
        return v ? static_cast<int>(strtol(v, nullptr, 0)) : 0;
    }();
    const uint32_t group = ((tune >> 8) & 0xFF) ? static_cast<uint32_t>((tune >> 8) & 0xFE) : 32u;
    const uint32_t wgs = (tune & 0xFF) ? static_cast<uint32_t>(tune & 0xFF) : 4u;
    const uint32_t n_groups = (a.n_rows + group - 1) / group;
    const uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>((n_groups + 3) / 4, static_cast<uint32_t>(a.n_cu) * wgs));
    const float4 *rows = static_cast<const float4 *>(a.rows);
#define RLR_MULTI(CHV, QV)                                                                                     \
    hipLaunchKernelGGL((scan_multi_kernel<CHV, QV>), dim3(blocks), dim3(256), 0, s, rows, a.query, q_pitch,    \
                       n_queries, a.scores, score_stride, a.n_rows, group)
#define RLR_MULTI_Q(CHV)                                  \
    do {                                                  \
        if (n_queries <= 2) RLR_MULTI(CHV, 2);            \
        else if (n_queries <= 4) RLR_MULTI(CHV, 4);       \
        else RLR_MULTI(CHV, 8);                           \
    } while (0)
    switch (ch) {
    case 1: RLR_MULTI_Q(1); break;
    case 2: RLR_MULTI_Q(2); break;
    case 3: RLR_MULTI_Q(3); break;
    default: RLR_MULTI_Q(4); break;
    }
#undef RLR_MULTI_Q
#undef RLR_MULTI
    *err = hipGetLastError();
    return true;
}

} // namespace rlr
