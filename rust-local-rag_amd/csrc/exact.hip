// exact.hip -- kernels that reproduce the reference's arithmetic bit for bit:
// strict left-to-right f32 accumulation, every product rounded before it is added, no
// FMA (this file is compiled with -ffp-contract=off and spells no fmaf), IEEE division
// and square root.  One lane owns one dot product, so the order inside a sum is the
// reference's; parallelism is across rows / pairs.
//
//   dot_ref        /root/reference/src/rag_engine.rs:1777-1779  (dot_product)
//   normalize      /root/reference/src/rag_engine.rs:1763-1771
//   mmr greedy     /root/reference/src/rag_engine.rs:788-835
#include "common.h"
#include "exact_dot.h"
#include "staged_dot.h"
#include "sort_emit.h"
#include "kernels.h"
#include "../../include/rlr_gpu.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>

namespace rlr {

namespace {

__device__ inline void stage_query(float *s_q, const float *__restrict__ query, uint32_t dim)
{
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x)
        s_q[i] = query[i];
    __syncthreads();
}

template <bool F16>
__global__ __launch_bounds__(64) void rescore_kernel(const float4 *__restrict__ rows, uint32_t pitch16,
                                                     uint32_t dim, const float *__restrict__ query,
                                                     const uint32_t *__restrict__ cand,
                                                     const SelectState *__restrict__ st,
                                                     uint64_t *__restrict__ packed_out, uint32_t n_pad)
{
    extern __shared__ __attribute__((aligned(16))) float s_q[];
    stage_query(s_q, query, dim);
    const uint32_t n = min(st->n_cand, st->cap);
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_pad)
        return;
    if (i >= n) {
        packed_out[i] = 0;
        return;
    }
    const uint32_t r = cand[i];
    const float s = dot_ref_row<F16>(rows + static_cast<size_t>(r) * pitch16, s_q, dim);
    packed_out[i] = pack_result(s, r);
}

template <bool F16>
__global__ __launch_bounds__(256) void rescore_staged_kernel(const float4 *__restrict__ rows, uint32_t pitch16,
                                                             uint32_t dim, const float *__restrict__ query,
                                                             const uint32_t *__restrict__ cand,
                                                             const SelectState *__restrict__ st,
                                                             uint64_t *__restrict__ packed_out, uint32_t cpb,
                                                             uint32_t *__restrict__ hist_clear)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    const uint32_t q_floats = (dim + 7) & ~7u;      // query, zero padded to a 16-byte unit of the row dtype
    float *s_q = s_mem;
    float *s_p = s_mem + q_floats;
    const uint32_t tid = threadIdx.x;
    const uint32_t gid = blockIdx.x * 256 + tid;
    if (hist_clear && gid < 2 * kHistBins)
        hist_clear[gid] = 0;
    const uint32_t n = min(st->n_cand, st->cap);
    const uint32_t base = blockIdx.x * cpb;
    if (base >= n)
        return;
    const uint32_t cnt = min(cpb, n - base);
    __shared__ uint32_t s_cand[16];
    for (uint32_t i = tid; i < q_floats; i += 256)
        s_q[i] = i < dim ? query[i] : 0.0f;
    if (tid < cnt)
        s_cand[tid] = cand[base + tid];
    __syncthreads();
    const float sc = staged_reference_dot<F16>(rows, pitch16, dim, s_q, s_p, s_cand, cnt, tid);
    if (tid < cnt)
        packed_out[base + tid] = pack_result(sc, s_cand[tid]);
}

// The same re-score for the batched path: blockIdx.y = query, the workgroups of a query stride over
// its guard band in groups of `cpb`; entries of band[q * band_stride ..] are packed (nominated score,
// row) and are overwritten in place with (reference-order score, row).  st[q].pad = band length.
template <bool F16>
__global__ __launch_bounds__(256) void batch_rescore_kernel(const float4 *__restrict__ rows, uint32_t pitch16,
                                                            uint32_t dim, const float *__restrict__ queries,
                                                            uint32_t q_pitch, uint64_t *__restrict__ band,
                                                            uint32_t band_stride, const SelectState *__restrict__ st,
                                                            uint32_t cpb)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    const uint32_t q_floats = (dim + 7) & ~7u;
    float *s_q = s_mem;
    float *s_p = s_mem + q_floats;
    __shared__ uint32_t s_cand[16];
    const uint32_t tid = threadIdx.x;
    const uint32_t q = blockIdx.y;
    const uint32_t n = st[q].pad;
    if (blockIdx.x * cpb >= n)
        return;
    const float *query = queries + static_cast<size_t>(q) * q_pitch;
    for (uint32_t i = tid; i < q_floats; i += 256)
        s_q[i] = i < dim ? query[i] : 0.0f;
    uint64_t *b = band + static_cast<size_t>(q) * band_stride;
    for (uint32_t base = blockIdx.x * cpb; base < n; base += gridDim.x * cpb) {
        const uint32_t cnt = min(cpb, n - base);
        __syncthreads(); // s_cand / s_p of the previous group are free (and s_q is staged)
        if (tid < cnt)
            s_cand[tid] = 0xFFFFFFFFu - static_cast<uint32_t>(b[base + tid] & 0xFFFFFFFFu);
        __syncthreads();
        const float sc = staged_reference_dot<F16>(rows, pitch16, dim, s_q, s_p, s_cand, cnt, tid);
        if (tid < cnt)
            b[base + tid] = pack_result(sc, s_cand[tid]);
    }
}

template <bool F16>
__global__ __launch_bounds__(64) void score_rows_kernel(const float4 *__restrict__ rows, uint32_t pitch16,
                                                        uint32_t dim, const float *__restrict__ query,
                                                        const uint32_t *__restrict__ list, uint32_t n,
                                                        float *__restrict__ out, const uint32_t *__restrict__ n_dev,
                                                        uint32_t n_rows_clamp)
{
    extern __shared__ __attribute__((aligned(16))) float s_q[];
    stage_query(s_q, query, dim);
    if (n_dev)
        n = min(n, *n_dev);
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t r = list[i] < n_rows_clamp ? list[i] : 0u;
    out[i] = dot_ref_row<F16>(rows + static_cast<size_t>(r) * pitch16, s_q, dim);
}

// The same scores with the rows staged through LDS (coalesced loads, products in parallel, then one reference-order
// chain per row): 8 rows per workgroup instead of 64 sequential lanes in one wave -- 1500 lexical rows of a hybrid
// search spread over 188 workgroups instead of 24 (24.5 -> 7.9 us at 768-d).
template <bool F16>
__global__ __launch_bounds__(256) void score_rows_staged_kernel(const float4 *__restrict__ rows, uint32_t pitch16, uint32_t dim,
                                                               const float *__restrict__ query,
                                                               const uint32_t *__restrict__ list, uint32_t n,
                                                               float *__restrict__ out, uint32_t cpb,
                                                               const uint32_t *__restrict__ n_dev, uint32_t n_rows_clamp)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    const uint32_t q_floats = (dim + 7) & ~7u;
    float *s_q = s_mem;
    float *s_p = s_mem + q_floats;
    __shared__ uint32_t s_cand[16];
    const uint32_t tid = threadIdx.x;
    const uint32_t base = blockIdx.x * cpb;
    if (n_dev)
        n = min(n, *n_dev);
    if (base >= n)
        return;
    const uint32_t cnt = min(cpb, n - base);
    for (uint32_t i = tid; i < q_floats; i += 256)
        s_q[i] = i < dim ? query[i] : 0.0f;
    if (tid < cnt) {
        const uint32_t r = list[base + tid];
        s_cand[tid] = r < n_rows_clamp ? r : 0u;
    }
    __syncthreads();
    const float sc = staged_reference_dot<F16>(rows, pitch16, dim, s_q, s_p, s_cand, cnt, tid);
    if (tid < cnt)
        out[base + tid] = sc;
}

// ---- normalize ---------------------------------------------------------------------
// phase 1: one lane per row, sequential sum of squares (reference order)
__global__ __launch_bounds__(64) void sumsq_kernel(const float *__restrict__ staging, uint32_t n,
                                                   uint32_t dim, float *__restrict__ sumsq)
{
    const uint32_t r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n)
        return;
    const float *v = staging + static_cast<size_t>(r) * dim;
    float s = 0.0f;
    for (uint32_t c = 0; c < dim; ++c) {
        float p = v[c] * v[c];
        s = s + p;
    }
    sumsq[r] = s;
}

// Correctly rounded sqrtf for positive normal x.  gfx950's v_sqrt_f32 is accurate to 1 ulp,
// not correctly rounded (and hipcc emits it bare for sqrtf/__fsqrt_rn), so the result is
// checked against the two neighbouring rounding boundaries in exact arithmetic: a boundary
// m = (y + y')/2 has <= 25 significant bits, so m*m is exact in binary64, and sqrt(x) can
// never sit exactly on a boundary (m*m needs more than 24 bits).
__device__ inline float sqrt_rn(float x)
{
    float y = __builtin_sqrtf(x);
    const double xd = static_cast<double>(x);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const uint32_t yb = __builtin_bit_cast(uint32_t, y);
        const float up = __builtin_bit_cast(float, yb + 1u);
        const float dn = __builtin_bit_cast(float, yb - 1u);
        const double m_hi = 0.5 * (static_cast<double>(y) + static_cast<double>(up));
        const double m_lo = 0.5 * (static_cast<double>(y) + static_cast<double>(dn));
        if (xd > m_hi * m_hi)
            y = up;
        else if (xd < m_lo * m_lo)
            y = dn;
    }
    return y;
}

__device__ inline float ref_scale(float v, float norm_sq, int do_normalize)
{
    // `if norm_sq > 1e-20 { x /= norm_sq.sqrt() }`  (rag_engine.rs:1765-1769)
    if (do_normalize && norm_sq > 1e-20f)
        return __fdiv_rn(v, sqrt_rn(norm_sq));
    return v;
}

// phase 2: coalesced scale + store in the index dtype; pad columns get 0
template <bool F16>
__global__ __launch_bounds__(256) void scale_store_kernel(const float *__restrict__ staging,
                                                          const float *__restrict__ sumsq, uint32_t n,
                                                          uint32_t dim, int do_normalize,
                                                          void *__restrict__ rows_out, uint32_t pitch16)
{
    const uint32_t pitch_e = F16 ? pitch16 * 8 : pitch16 * 4;
    const size_t total = static_cast<size_t>(n) * pitch_e;
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += stride) {
        const uint32_t r = static_cast<uint32_t>(i / pitch_e);
        const uint32_t c = static_cast<uint32_t>(i - static_cast<size_t>(r) * pitch_e);
        float v = 0.0f;
        if (c < dim)
            v = ref_scale(staging[static_cast<size_t>(r) * dim + c], do_normalize ? sumsq[r] : 0.0f,
                          do_normalize);
        if constexpr (F16)
            static_cast<_Float16 *>(rows_out)[i] = static_cast<_Float16>(v);
        else
            static_cast<float *>(rows_out)[i] = v;
    }
}

// ---- synthetic corpus (twin of oracle/rlr_oracle.c: rlr_o_synth_raw) -----------------
__device__ inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__device__ inline int32_t ih4(uint64_t h)
{
    return static_cast<int32_t>((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48)) - 131070;
}

__device__ inline float synth_raw(uint64_t s, uint64_t row, uint32_t col, uint32_t d, uint32_t n_clusters)
{
    const uint64_t idx = row * static_cast<uint64_t>(d) + col;
    int32_t t = ih4(mix64(s + (idx + 1) * 0x9E3779B97F4A7C15ULL));
    const bool tight = (n_clusters & 0x80000000u) != 0; // near-copies inside a cluster (oracle/rlr_oracle.c: rlr_o_synth_raw)
    n_clusters &= 0x7FFFFFFFu;
    if (n_clusters) {
        const uint64_t cl = mix64(s ^ (row + 0x632BE59BD9B4E019ULL)) % n_clusters;
        const uint64_t cidx = cl * static_cast<uint64_t>(d) + col;
        const int32_t c = ih4(mix64((s ^ 0xC1A57E55C1A57E55ULL) + (cidx + 1) * 0x9E3779B97F4A7C15ULL));
        if (tight)
            t = t / 16;
        t += 2 * c;
    }
    return static_cast<float>(t) * (1.0f / 65536.0f);
}

__global__ __launch_bounds__(64) void synth_sumsq_kernel(uint64_t s, uint64_t row0, uint32_t n, uint32_t dim,
                                                         uint32_t n_clusters, float *__restrict__ sumsq)
{
    const uint32_t r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n)
        return;
    float acc = 0.0f;
    for (uint32_t c = 0; c < dim; ++c) {
        const float v = synth_raw(s, row0 + r, c, dim, n_clusters);
        float p = v * v;
        acc = acc + p;
    }
    sumsq[r] = acc;
}

template <bool F16>
__global__ __launch_bounds__(256) void synth_store_kernel(uint64_t s, uint64_t row0, uint32_t n, uint32_t dim,
                                                          uint32_t n_clusters, const float *__restrict__ sumsq,
                                                          void *__restrict__ rows_out, uint32_t pitch16)
{
    const uint32_t pitch_e = F16 ? pitch16 * 8 : pitch16 * 4;
    const size_t total = static_cast<size_t>(n) * pitch_e;
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += stride) {
        const uint32_t r = static_cast<uint32_t>(i / pitch_e);
        const uint32_t c = static_cast<uint32_t>(i - static_cast<size_t>(r) * pitch_e);
        float v = 0.0f;
        if (c < dim)
            v = ref_scale(synth_raw(s, row0 + r, c, dim, n_clusters), sumsq[r], 1);
        if constexpr (F16)
            static_cast<_Float16 *>(rows_out)[i] = static_cast<_Float16>(v);
        else
            static_cast<float *>(rows_out)[i] = v;
    }
}

// ---- row movement ------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void gather_f32_kernel(const void *__restrict__ rows, uint32_t pitch16,
                                                         uint32_t dim, const uint32_t *__restrict__ list,
                                                         uint32_t n, float *__restrict__ out)
{
    const size_t total = static_cast<size_t>(n) * dim;
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += stride) {
        const uint32_t k = static_cast<uint32_t>(i / dim);
        const uint32_t c = static_cast<uint32_t>(i - static_cast<size_t>(k) * dim);
        const size_t r = list[k];
        if constexpr (F16)
            out[i] = h2f(static_cast<const uint16_t *>(rows)[r * pitch16 * 8 + c]);
        else
            out[i] = static_cast<const float *>(rows)[r * pitch16 * 4 + c];
    }
}

__global__ __launch_bounds__(256) void compact_rows_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst,
                                                           uint32_t pitch16, const uint32_t *__restrict__ keep,
                                                           uint32_t n_keep)
{
    const size_t total = static_cast<size_t>(n_keep) * pitch16;
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += stride) {
        const uint32_t k = static_cast<uint32_t>(i / pitch16);
        const uint32_t u = static_cast<uint32_t>(i - static_cast<size_t>(k) * pitch16);
        dst[i] = src[static_cast<size_t>(keep[k]) * pitch16 + u];
    }
}

// ---- MMR ----------------------------------------------------------------------------
__device__ inline bool finite_f(float x)
{
    return (__builtin_bit_cast(uint32_t, x) & 0x7F800000u) != 0x7F800000u;
}

// What the Gram kernels store: the dot product, or -inf where it is not finite.  The greedy loops read a similarity only
// as `if sim.is_finite() { max_sim = max_sim.max(sim) }` on a max_sim that starts at +0.0, so -inf ("never raises it") stands
// for every non-finite value and the chain compares without a class test.
__device__ inline float gram_entry(float dot)
{
    return finite_f(dot) ? dot : -__builtin_inff();
}

// gram[i][j] = dot_ref(pool_i, pool_j), one lane per pair (j <= i computed, mirrored):
// a*b is commutative and the summation order is the same, so dot(i,j) == dot(j,i) bitwise.
__global__ __launch_bounds__(64) void gram_kernel(const float *__restrict__ pool, uint32_t P, uint32_t dim,
                                                  float *__restrict__ gram)
{
    // blockIdx.z = query of a batch (pool and gram are P-strided per query)
    pool += static_cast<size_t>(blockIdx.z) * P * dim;
    gram += static_cast<size_t>(blockIdx.z) * P * P;
    const uint32_t i = blockIdx.y;
    const uint32_t j = blockIdx.x * 64 + threadIdx.x;
    if (j > i || i >= P)
        return;
    const float *a = pool + static_cast<size_t>(i) * dim;
    const float *b = pool + static_cast<size_t>(j) * dim;
    float s = 0.0f;
    uint32_t c = 0;
    if ((dim & 3u) == 0) {
        const float4 *a4 = reinterpret_cast<const float4 *>(a);
        const float4 *b4 = reinterpret_cast<const float4 *>(b);
        for (; c + 16 <= dim; c += 16) {
            float4 x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x[u] = a4[c / 4 + u];
                y[u] = b4[c / 4 + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float p;
                p = x[u].x * y[u].x; s = s + p;
                p = x[u].y * y[u].y; s = s + p;
                p = x[u].z * y[u].z; s = s + p;
                p = x[u].w * y[u].w; s = s + p;
            }
        }
    }
    for (; c < dim; ++c) {
        float p = a[c] * b[c];
        s = s + p;
    }
    s = gram_entry(s);
    gram[static_cast<size_t>(i) * P + j] = s;
    gram[static_cast<size_t>(j) * P + i] = s;
}

// One staged float4 of a pool row (the vector path: dim % 4 == 0): rows as dense f32 (SRC 0, row = pool index), through the
// index (SRC 1 f32, SRC 2 binary16 widened exactly; row = index row).  Unconditional -- the caller clamps row and column
// into range and discards what it did not want -- so that a thread's loads are all in flight together: behind a
// per-element `if` hipcc waited for every row-list entry and every row fragment one after the other.
template <int SRC>
__device__ inline float4 gram_load4(const float *pool, const void *rows, uint32_t pitch16, uint32_t dim, uint32_t row, uint32_t col)
{
    if constexpr (SRC == 2) {
        const uint2 h = *reinterpret_cast<const uint2 *>(static_cast<const uint16_t *>(rows) + static_cast<size_t>(row) * pitch16 * 8 + col);
        return make_float4(h2f(static_cast<uint16_t>(h.x)), h2f(static_cast<uint16_t>(h.x >> 16)), h2f(static_cast<uint16_t>(h.y)),
                           h2f(static_cast<uint16_t>(h.y >> 16)));
    } else if constexpr (SRC == 1) {
        return *reinterpret_cast<const float4 *>(static_cast<const float *>(rows) + static_cast<size_t>(row) * pitch16 * 4 + col);
    } else {
        return *reinterpret_cast<const float4 *>(pool + static_cast<size_t>(row) * dim + col);
    }
}

// Tiled Gram: a workgroup owns a 32 x 32 block of pairs (lower triangle of blocks only), stages
// 64-column chunks of the 32 + 32 rows in LDS with coalesced loads, and every thread carries four
// pairs (rows {ty, ty+16} x {tx, tx+16}: conflict-free LDS rows at a 68-float pitch) through the
// chunks in ascending k, so each sum is still the reference's strict left-to-right chain.  Reads
// each pool row 2 x ceil(P/32) times from L2 instead of P times: 27 -> ~8 us per 300-row pool and
// ~5x on the batched MMR.

// R = side of the per-thread register tile of pairs: a workgroup (16 x 16 threads) owns a (16 R) x (16 R) block.
// R = 1 (16 x 16 blocks, 210 of them for a 300-row pool) fills the chip for a single query; R = 2 (32 x 32) for larger
// single pools; R = 4 (64 x 64) halves the LDS reads per multiply (8 float4 per 64 products instead of 4 per 16) and is
// used for batches, where there are thousands of blocks anyway (launch_gram_src picks).  GK = columns per staged chunk.
// SRC 0: `pool` is a dense P x dim f32 matrix per query.  SRC 1 / 2: the pool rows are read straight from the index
// (f32 / binary16 rows, widened exactly) through the query's row list -- no gathered f32 copy in between: config 5's
// share used to write and re-read 1.26 GB of it per 1024 pools.
template <int R, int SRC, int GK>
__global__ __launch_bounds__(256) void gram_tiled_kernel(const float *__restrict__ pool, uint32_t P, uint32_t dim,
                                                         float *__restrict__ gram, const void *__restrict__ rows,
                                                         uint32_t pitch16, const uint32_t *__restrict__ list)
{
    constexpr int kGT = 16 * R; // pairs per block side
    constexpr int kGK = GK, kGPitch = GK + 4; // columns per staged chunk; row pitch in floats (pad: no bank conflicts)
    __shared__ __attribute__((aligned(16))) float sa[kGT * kGPitch];
    __shared__ __attribute__((aligned(16))) float sb[kGT * kGPitch];
    if constexpr (SRC == 0)
        pool += static_cast<size_t>(blockIdx.z) * P * dim;
    else
        list += static_cast<size_t>(blockIdx.z) * P;
    gram += static_cast<size_t>(blockIdx.z) * P * P;
    // linear block id -> (bi >= bj) in the lower triangle
    uint32_t bi = 0, rem = blockIdx.x;
    while (rem > bi) {
        rem -= bi + 1;
        ++bi;
    }
    const uint32_t bj = rem;
    const uint32_t i0 = bi * kGT, j0 = bj * kGT;
    const uint32_t t = threadIdx.x, ty = t >> 4, tx = t & 15;
    float acc[R][R];
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
        for (int b = 0; b < R; ++b)
            acc[a][b] = 0.0f;
    const bool vec = (dim & 3u) == 0;
    // the next chunk's rows travel global -> registers while the current chunk is multiplied out of LDS (one round of
    // load latency per chunk used to sit between the two barriers)
    constexpr int kPre = 2 * kGT * (kGK / 4) / 256; // float4 per thread per chunk
    float4 pre[kPre];
    // where thread t's u-th staged float4 comes from: the same row for every chunk (index rows looked up once)
    uint32_t src_row[kPre];
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
        const uint32_t r = (t + 256 * u) / (kGK / 4);
        const uint32_t row = min(r >= kGT ? j0 + (r - kGT) : i0 + r, P - 1);
        src_row[u] = SRC == 0 ? row : list[row];
    }
    auto fetch = [&](uint32_t k0) {
        const uint32_t kc = min(static_cast<uint32_t>(kGK), dim - k0);
        if (vec) {
#pragma unroll
            for (int u = 0; u < kPre; ++u) {
                const uint32_t idx = t + 256 * u;
                const uint32_t r = idx / (kGK / 4), c4 = idx % (kGK / 4);
                const uint32_t row = r >= kGT ? j0 + (r - kGT) : i0 + r;
                const float4 v = gram_load4<SRC>(pool, rows, pitch16, dim, src_row[u], min(k0 + c4 * 4, dim - 4));
                pre[u] = (row < P && c4 * 4 < kc) ? v : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const uint32_t idx = t + 256 * u;
            const uint32_t r = idx / (kGK / 4), c4 = idx % (kGK / 4);
            const uint32_t row = r >= kGT ? j0 + (r - kGT) : i0 + r;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (row < P && c4 * 4 < kc) {
                const uint32_t col = k0 + c4 * 4;
                if constexpr (SRC == 2) {
                    const uint16_t *src = static_cast<const uint16_t *>(rows) + static_cast<size_t>(src_row[u]) * pitch16 * 8 + col;
                    v.x = h2f(src[0]);
                    if (c4 * 4 + 1 < kc) v.y = h2f(src[1]);
                    if (c4 * 4 + 2 < kc) v.z = h2f(src[2]);
                    if (c4 * 4 + 3 < kc) v.w = h2f(src[3]);
                } else {
                    const float *src = SRC == 0 ? pool + static_cast<size_t>(row) * dim + col
                                                : static_cast<const float *>(rows) + static_cast<size_t>(src_row[u]) * pitch16 * 4 + col;
                    v.x = src[0];
                    if (c4 * 4 + 1 < kc) v.y = src[1];
                    if (c4 * 4 + 2 < kc) v.z = src[2];
                    if (c4 * 4 + 3 < kc) v.w = src[3];
                }
            }
            pre[u] = v;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const uint32_t idx = t + 256 * u;
            const uint32_t r = idx / (kGK / 4), c4 = idx % (kGK / 4);
            float *dst = (r >= kGT ? sb + (r - kGT) * kGPitch : sa + r * kGPitch) + c4 * 4;
            *reinterpret_cast<float4 *>(dst) = pre[u];
        }
    };
    fetch(0);
    stash();
    __syncthreads();
    for (uint32_t k0 = 0; k0 < dim; k0 += kGK) {
        const uint32_t kc = min(static_cast<uint32_t>(kGK), dim - k0);
        const bool more = k0 + kGK < dim;
        if (more)
            fetch(k0 + kGK);
        const uint32_t n4 = kc / 4;
        // every sum is still the reference's strict left-to-right chain: k ascends, product rounded, then added.
        // (one pair per thread is one wave per SIMD for a single pool: unrolled, the LDS reads of eight steps in flight together)
        constexpr int kUnroll = R == 1 ? 8 : 1;
#pragma unroll kUnroll
        for (uint32_t c = 0; c < n4; ++c) {
            float4 xa[R], yb[R];
#pragma unroll
            for (int a = 0; a < R; ++a)
                xa[a] = reinterpret_cast<const float4 *>(sa + (ty + 16 * a) * kGPitch)[c];
#pragma unroll
            for (int b = 0; b < R; ++b)
                yb[b] = reinterpret_cast<const float4 *>(sb + (tx + 16 * b) * kGPitch)[c];
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int b = 0; b < R; ++b) {
                    float p;
                    p = xa[a].x * yb[b].x; acc[a][b] = acc[a][b] + p;
                    p = xa[a].y * yb[b].y; acc[a][b] = acc[a][b] + p;
                    p = xa[a].z * yb[b].z; acc[a][b] = acc[a][b] + p;
                    p = xa[a].w * yb[b].w; acc[a][b] = acc[a][b] + p;
                }
        }
        for (uint32_t e = n4 * 4; e < kc; ++e) { // dim % 4 tail
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int b = 0; b < R; ++b) {
                    const float p = sa[(ty + 16 * a) * kGPitch + e] * sb[(tx + 16 * b) * kGPitch + e];
                    acc[a][b] = acc[a][b] + p;
                }
        }
        __syncthreads();
        if (more) {
            stash();
            __syncthreads();
        }
    }
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
        for (int b = 0; b < R; ++b) {
            const uint32_t ri = i0 + ty + 16 * a, cj = j0 + tx + 16 * b;
            if (ri < P && cj < P) {
                const float g = gram_entry(acc[a][b]);
                gram[static_cast<size_t>(ri) * P + cj] = g;
                gram[static_cast<size_t>(cj) * P + ri] = g; // dot(i,j) == dot(j,i) bitwise
            }
        }
}

// Greedy selection by one wavefront; `rem` replays the reference's Vec::swap_remove
// bookkeeping so the visiting order (hence tie-breaking under strict `>`) is identical.
__global__ __launch_bounds__(64) void mmr_greedy_kernel(const float *__restrict__ gram,
                                                        const float *__restrict__ scores, uint32_t P,
                                                        uint32_t k, float lambda, uint32_t *__restrict__ out_order,
                                                        float *__restrict__ out_mmr, uint32_t *__restrict__ out_n)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    uint32_t *rem = reinterpret_cast<uint32_t *>(s_raw);          // P: position -> candidate
    float *max_sim = reinterpret_cast<float *>(s_raw) + P;        // P: per candidate, fold(0.0, max)
    float *rel = reinterpret_cast<float *>(s_raw) + 2 * P;        // P: relevance
    const uint32_t lane = threadIdx.x;
    if (P == 0) {
        if (lane == 0)
            *out_n = 0;
        return;
    }
    for (uint32_t p = lane; p < P; p += 64) {
        rem[p] = p;
        max_sim[p] = 0.0f;
        rel[p] = scores[p];
    }
    __syncthreads();
    // selected.push(remaining.swap_remove(0))  -- unconditional (rag_engine.rs:782-785)
    uint32_t n_rem = P, n_sel = 0;
    uint32_t last = rem[0];
    __syncthreads();
    if (lane == 0) {
        out_order[0] = last;
        out_mmr[0] = __builtin_bit_cast(float, 0x7FC00000u);
        rem[0] = rem[P - 1];
    }
    n_sel = 1;
    n_rem = P - 1;
    __syncthreads();

    const float one_minus = 1.0f - lambda;
    while (n_sel < k && n_rem > 0) {
        const float *g_last = gram + static_cast<size_t>(last) * P; // row `last` == column `last`
        uint64_t best = 0; // (key(mmr) << 32) | ~position ; 0 = no finite candidate
        for (uint32_t p = lane; p < n_rem; p += 64) {
            const uint32_t c = rem[p];
            const float sim = g_last[c];
            float ms = max_sim[c];
            if (finite_f(sim))
                ms = fmaxf(ms, sim);
            max_sim[c] = ms;
            const float r = rel[c];
            if (!finite_f(r))
                continue;
            const float t0 = one_minus * r;
            const float t1 = lambda * ms;
            float m = t0 - t1;
            if (!finite_f(m))
                continue;
            if (m == 0.0f)
                m = 0.0f; // -0 and +0 compare equal in the reference
            const uint64_t cand = (static_cast<uint64_t>(score_key(m)) << 32) | (0xFFFFFFFFu - p);
            if (cand > best)
                best = cand;
        }
        // wavefront argmax: larger mmr wins, equal mmr -> lower position (first visited)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t hi = __shfl_xor(static_cast<uint32_t>(best >> 32), off);
            const uint32_t lo = __shfl_xor(static_cast<uint32_t>(best & 0xFFFFFFFFu), off);
            const uint64_t other = (static_cast<uint64_t>(hi) << 32) | lo;
            if (other > best)
                best = other;
        }
        if (best == 0) // `if best_mmr_score == NEG_INFINITY { break }`
            break;
        const uint32_t best_p = 0xFFFFFFFFu - static_cast<uint32_t>(best & 0xFFFFFFFFu);
        __syncthreads();
        last = rem[best_p];
        __syncthreads();
        if (lane == 0) {
            out_order[n_sel] = last;
            // the logged value, bit for bit (the compare above ran on the -0 -> +0 canonical form)
            out_mmr[n_sel] = one_minus * rel[last] - lambda * max_sim[last];
            rem[best_p] = rem[n_rem - 1]; // swap_remove(best_idx)
        }
        n_sel++;
        n_rem--;
        __syncthreads();
    }
    if (lane == 0)
        *out_n = n_sel;
}

// The result block of the fused search -> MMR paths, written by ONE wave (sort_emit.h: [row | cos | combined | lexical] x
// k_cap, n, status, checksum, done): pick i is pool slot order(i).  The values, then n / status / the checksum of it all, then --
// behind a system-scope fence -- the completion word a polling host waits for (the word can still overtake the values on
// their way through PCIe; the checksum settles that on the host).
template <typename Order>
__device__ __forceinline__ void emit_result_block(const MmrEmit &emit, uint32_t n_sel, uint32_t lane, Order order)
{
    const uint32_t status = emit.info[1];
    const uint32_t n = status ? 0u : min(n_sel, emit.k_cap);
    uint32_t chk = 0;
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t o = order(i);
        const uint32_t w0 = emit.list[o], w1 = __builtin_bit_cast(uint32_t, emit.cosv[o]);
        const uint32_t w2 = __builtin_bit_cast(uint32_t, emit.comb[o]);
        const uint32_t w3 = emit.lexv ? __builtin_bit_cast(uint32_t, emit.lexv[o]) : 0u;
        emit.h_out[i] = w0;
        emit.h_out[emit.k_cap + i] = w1;
        emit.h_out[2 * emit.k_cap + i] = w2;
        emit.h_out[3 * emit.k_cap + i] = w3;
        chk += result_chk_term(w0, i) + result_chk_term(w1, emit.k_cap + i) + result_chk_term(w2, 2 * emit.k_cap + i) +
               result_chk_term(w3, 3 * emit.k_cap + i);
    }
    __threadfence_system();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        chk += static_cast<uint32_t>(__shfl_xor(static_cast<int>(chk), off));
    if (lane == 0) {
        emit.h_out[4 * emit.k_cap] = n;
        emit.h_out[4 * emit.k_cap + 1] = status;
        emit.h_out[4 * emit.k_cap + 2] = block_chk_tail(chk, n, status, emit.k_cap);
        __threadfence_system();
        emit.h_out[4 * emit.k_cap + 3] = kBlockDone;
    }
}

// an unusable pool (size 0, its status in info[1]): the same block with no values; one thread
__device__ __forceinline__ void emit_empty_block(const MmrEmit &emit)
{
    const uint32_t status = emit.info[1];
    emit.h_out[4 * emit.k_cap] = 0;
    emit.h_out[4 * emit.k_cap + 1] = status;
    emit.h_out[4 * emit.k_cap + 2] = block_chk_tail(0u, 0u, status, emit.k_cap);
    __threadfence_system();
    emit.h_out[4 * emit.k_cap + 3] = kBlockDone;
}

// Register-resident greedy MMR for pools of <= 64*J candidates: lane l owns candidates l, l+64, ...
// with their relevance, running max-similarity and current position in the reference's
// `remaining` vector held in VGPRs.  A step is one batch of independent L2 loads of the last
// pick's Gram row, a handful of VALU ops, a DPP wavefront max over the MMR values and a DPP
// wavefront min over the positions of the lanes that hold that max ("first in visiting order
// wins" under the reference's strict `>`), and the swap_remove position update -- no LDS, no
// barriers.  ~0.3 us per pick instead of ~2.2 us for the LDS version.
template <int J>
__global__ __launch_bounds__(256) void mmr_greedy_reg_kernel(const float *__restrict__ gram,
                                                             const float *__restrict__ scores, uint32_t P, uint32_t k,
                                                             float lambda, uint32_t *__restrict__ out_order,
                                                             float *__restrict__ out_mmr, uint32_t *__restrict__ out_n,
                                                             const uint32_t *__restrict__ sizes, MmrEmit emit)
{
    // batch: blockIdx.x = query; arrays are strided by the launch-wide P, the pool size is sizes[q]
    {
        const uint32_t stride = P;
        gram += static_cast<size_t>(blockIdx.x) * stride * stride;
        scores += static_cast<size_t>(blockIdx.x) * stride;
        out_order += static_cast<size_t>(blockIdx.x) * stride;
        out_mmr += static_cast<size_t>(blockIdx.x) * stride;
        out_n += blockIdx.x;
    }
    const uint32_t g_stride = P;
    if (sizes)
        P = sizes[blockIdx.x];
    if (P == 0) {
        if (threadIdx.x == 0) {
            *out_n = 0;
            if (emit.h_out) // (an unusable pool arrives here as size 0 with its status in info[1])
                emit_empty_block(emit);
        }
        return;
    }
    // The Gram matrix was just written by other CUs (possibly other XCDs): its first touch from
    // this CU is an Infinity-Cache/HBM miss (~0.4 us), and every pick reads a different row, so
    // the greedy chain would pay that miss 99 times.  All four waves first sweep the matrix
    // (P*P*4 bytes, 360 KB at P = 300) into this XCD's L2; the chain's loads then hit L2.
    {
        const float4 *g4 = reinterpret_cast<const float4 *>(gram);
        const uint32_t n4 = (P * g_stride) / 4;
        float warm = 0.0f;
        for (uint32_t i = threadIdx.x * 8; i < n4; i += 256 * 8) // one 128-B line per thread and step
            warm += g4[i].x;
        asm volatile("" ::"v"(warm));
    }
    __syncthreads();
    if (threadIdx.x >= 64)
        return;
    const uint32_t lane = threadIdx.x;
    // The loop body is written branch-free (selects, no `continue`): with branches hipcc builds a saveexec / branch
    // ladder per candidate, and one wave alone on its SIMD pays every one of those at ~5 cycles per instruction -- the
    // chain took 1.07 us per pick, about two thirds of it control flow.  There are no per-candidate flags either
    // (hipcc packs bool arrays into bytes and unpacks them every pick): a slot that is empty or already picked carries
    // t0 = NaN, so its MMR value is never finite and never a candidate, and position ~0, which no live position
    // equals; a relevance that is not finite makes t0, hence the MMR value, non-finite by itself (`rel.is_finite()`).
    float t0[J], ms[J];          // (1 - lambda) * relevance (loop invariant; NaN = not selectable), running max similarity
    uint32_t pos[J], idx[J];     // position in the reference's `remaining` (~0 = gone); clamped Gram column of the candidate
    const float one_minus = 1.0f - lambda;
    const float nan_f = __builtin_bit_cast(float, 0x7FC00000u);
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const uint32_t c = lane + 64 * j;
        idx[j] = min(c, P - 1);
        const float r = scores[idx[j]];
        // selected.push(remaining.swap_remove(0)): candidate 0 goes first, the last one takes slot 0
        const bool usable = (c < P) & (c != 0);
        t0[j] = usable ? one_minus * r : nan_f;
        ms[j] = 0.0f;
        pos[j] = usable ? ((c == P - 1) ? 0u : c) : 0xFFFFFFFFu;
    }
    uint32_t n_rem = P - 1, n_sel = 1, last = 0;
    if (lane == 0) {
        out_order[0] = 0;
        out_mmr[0] = nan_f;
    }
    const float neg_inf = -__builtin_inff();
    while (n_sel < k && n_rem > 0) {
        const float *g_last = gram + static_cast<size_t>(last) * g_stride;
        float sim[J];
#pragma unroll
        for (int j = 0; j < J; ++j)
            sim[j] = g_last[idx[j]]; // empty slots load a valid (clamped) column and ignore it
        float best_m = neg_inf;
        uint32_t best_pos = 0xFFFFFFFFu;
        float raw[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            // `if sim.is_finite() { max_sim = max_sim.max(sim) }` on a max_sim that starts at +0.0 and is never NaN
            const bool raise = finite_f(sim[j]) & (sim[j] > ms[j]);
            ms[j] = raise ? sim[j] : ms[j];
            const float t1 = lambda * ms[j];
            const float m0 = t0[j] - t1;
            raw[j] = m0;                            // what the reference logs for the winner (sign of zero included)
            // (-0 and +0 compare equal in the reference, and so they do in the float compares below)
            const bool better = finite_f(m0) & ((m0 > best_m) | ((m0 == best_m) & (pos[j] < best_pos)));
            best_m = better ? m0 : best_m;
            best_pos = better ? pos[j] : best_pos;
        }
        const float wm = wave_max_f32_no_nan(best_m); // best_m is -inf or a finite MMR value
        if (wm == neg_inf) // no finite candidate left
            break;
        const uint32_t wp = wave_min_u32(best_m == wm ? best_pos : 0xFFFFFFFFu);
        uint32_t win = 0;
        float win_raw = 0.0f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const bool hit = pos[j] == wp;
            win = hit ? lane + 64 * j + 1 : win;
            win_raw = hit ? raw[j] : win_raw;
            t0[j] = hit ? nan_f : t0[j];
            // swap_remove(best_idx): the winner leaves, the candidate in the last slot moves into the freed one
            pos[j] = hit ? 0xFFFFFFFFu : (pos[j] == n_rem - 1 ? wp : pos[j]);
        }
        // exactly one lane holds the winner: broadcast its candidate index
        const unsigned long long ball = __ballot(win != 0);
        const int src = __builtin_ctzll(ball);
        last = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(win), src)) - 1;
        const float wm_raw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, win_raw), src));
        if (lane == 0) {
            out_order[n_sel] = last;
            out_mmr[n_sel] = wm_raw;
        }
        n_sel++;
        n_rem--;
    }
    if (lane == 0)
        *out_n = n_sel;
    if (emit.h_out) { // the picks straight into the caller's (pinned) result block
        __threadfence(); // lane 0's out_order stores, read back by all lanes
        emit_result_block(emit, n_sel, lane, [&](uint32_t i) { return out_order[i]; });
    }
}

// max over a lane's J values that skips NaN, never below -inf: v_max3_f32 returns the largest non-NaN operand (all values
// here are quiet NaNs or numbers), a tree two operands wide per step instead of a compare-and-select chain
template <int J>
__device__ inline float lane_max_skip_nan(const float (&v)[J])
{
    float best = -__builtin_inff();
    int j = 0;
#pragma unroll
    for (; j + 1 < J; j += 2)
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(best) : "v"(best), "v"(v[j]), "v"(v[j + 1]));
    if (j < J)
        asm("v_max_f32 %0, %1, %2" : "=v"(best) : "v"(best), "v"(v[j]));
    return best;
}

// The same chain with the tie-break taken off the common path.
//
// A pick of the register-resident kernel above is ~190 instructions of one wave alone on its SIMD -- an issue slot every
// four cycles whatever the instruction, eight when it depends on the one before -- plus the L2 round trip of its Gram-row
// loads: 0.60 us.  About a third of those instructions keep every candidate's position in the reference's `remaining`
// vector (swap_remove moves the last entry into the freed slot) and run a second wavefront reduction over positions --
// needed only when two candidates hold the same maximal MMR value ("first in visiting order wins" under the strict `>`).
// Here the wave counts the holders of the maximum (one compare and one ballot per slot); a single holder is the pick, and
// only a tie replays the removals logged so far on a copy of `remaining` in LDS (lane 0, incrementally: each removal is
// replayed once) and takes the lowest position among the tied.  Non-finite values never reach the compares: a relevance
// that makes (1 - lambda) * rel non-finite is a NaN slot from the start, the Gram kernels store -inf for a non-finite
// similarity (gram_entry), the running maximum ignores NaN and -inf by itself, and an MMR value of +inf (overflow) sends
// that pick through a masked recount.  The logged value is the maximum itself unless it is a zero, whose sign is read
// from the winner.  Picks and logged values collect in LDS and leave in one burst: the chain holds no global stores.
// 0.47 us per pick (52 us for 99 picks of 300 against 64.5).  [Also tried: the packed upper triangle of the matrix in
// LDS, filled by sixteen waves -- 0.44 us per pick, but 3 us more in front of the chain: no gain, dropped.]
template <int J>
__global__ __launch_bounds__(256) void mmr_greedy_lazy_kernel(const float *__restrict__ gram, const float *__restrict__ scores,
                                                              uint32_t P, uint32_t k, float lambda,
                                                              uint32_t *__restrict__ out_order, float *__restrict__ out_mmr,
                                                              uint32_t *__restrict__ out_n, const uint32_t *__restrict__ sizes,
                                                              MmrEmit emit)
{
    constexpr uint32_t kSlots = 64 * J;
    __shared__ uint2 s_log[kSlots];                // pick i: candidate, logged MMR value (bits)
    __shared__ uint16_t s_rem[kSlots], s_posof[kSlots];
    {
        const uint32_t stride = P;
        gram += static_cast<size_t>(blockIdx.x) * stride * stride;
        scores += static_cast<size_t>(blockIdx.x) * stride;
        out_order += static_cast<size_t>(blockIdx.x) * stride;
        out_mmr += static_cast<size_t>(blockIdx.x) * stride;
        out_n += blockIdx.x;
    }
    const uint32_t g_stride = P;
    if (sizes)
        P = sizes[blockIdx.x];
    if (P == 0) {
        if (threadIdx.x == 0) {
            *out_n = 0;
            if (emit.h_out) // (an unusable pool arrives here as size 0 with its status in info[1])
                emit_empty_block(emit);
        }
        return;
    }
    // (the L2 warm-up of mmr_greedy_reg_kernel: every pick reads a different row of a matrix other CUs just wrote)
    {
        const float4 *g4 = reinterpret_cast<const float4 *>(gram);
        const uint32_t n4 = (P * g_stride) / 4;
        float warm = 0.0f;
        for (uint32_t i = threadIdx.x * 8; i < n4; i += 256 * 8) // one 128-B line per thread and step
            warm += g4[i].x;
        asm volatile("" ::"v"(warm));
    }
    __syncthreads();
    if (threadIdx.x >= 64)
        return;
    const uint32_t lane = threadIdx.x;
    for (uint32_t c = lane; c < P; c += 64) { // the reference's `remaining` before any removal
        s_rem[c] = static_cast<uint16_t>(c);
        s_posof[c] = static_cast<uint16_t>(c);
    }
    float t0[J], ms[J];  // (1 - lambda) * relevance (loop invariant; NaN = not selectable), running max similarity
    uint32_t idx[J];     // clamped Gram column of the candidate
    const float one_minus = 1.0f - lambda;
    const float nan_f = __builtin_bit_cast(float, 0x7FC00000u);
    const float neg_inf = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const uint32_t c = lane + 64 * j;
        idx[j] = min(c, P - 1);
        const float t = one_minus * scores[idx[j]];
        // selected.push(remaining.swap_remove(0)): candidate 0 goes first
        t0[j] = ((c < P) & (c != 0) & finite_f(t)) ? t : nan_f;
        ms[j] = 0.0f;
    }
    uint32_t n_sel = 1, last = 0, replayed = 0;
    s_log[0] = make_uint2(0u, 0x7FC00000u);
    while (n_sel < k && n_sel < P) {
        const float *g_last = gram + static_cast<size_t>(last) * g_stride; // row `last` == column `last`
        float sim[J];
#pragma unroll
        for (int j = 0; j < J; ++j)
            sim[j] = g_last[idx[j]]; // empty slots load a valid (clamped) column and ignore it
        float m0[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            ms[j] = sim[j] > ms[j] ? sim[j] : ms[j]; // `if sim.is_finite() { max_sim = max_sim.max(sim) }`: false for -inf
            const float t1 = lambda * ms[j];
            m0[j] = t0[j] - t1;
        }
        float wm = wave_max_f32_no_nan(lane_max_skip_nan<J>(m0));
        if (wm == __builtin_inff()) { // an overflowed value is no candidate (`mmr_score.is_finite()`): recount without those
#pragma unroll
            for (int j = 0; j < J; ++j)
                m0[j] = finite_f(m0[j]) ? m0[j] : nan_f;
            wm = wave_max_f32_no_nan(lane_max_skip_nan<J>(m0));
        }
        if (wm == neg_inf) // no finite candidate left
            break;
        // the tail of a pick, once per way of finding the winner (a flag array merged from two branches would be packed
        // into bytes and unpacked again by hipcc, every pick)
        auto take = [&](const bool (&hit)[J]) {
            uint32_t win = 0;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                win = hit[j] ? lane + 64 * j + 1 : win;
                t0[j] = hit[j] ? nan_f : t0[j];
            }
            // exactly one lane holds the winner: broadcast its candidate index
            const int src = __builtin_ctzll(__ballot(win != 0));
            last = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(win), src)) - 1;
            float wm_raw = wm;
            if (wm == 0.0f) { // the reference logs the winner's own zero, sign included
                float z = 0.0f;
#pragma unroll
                for (int j = 0; j < J; ++j)
                    z = hit[j] ? m0[j] : z;
                wm_raw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), src));
            }
            s_log[n_sel] = make_uint2(last, __builtin_bit_cast(uint32_t, wm_raw)); // (every lane, the same words)
            // lane 0 reads the log back in the tie path and every lane at the end: ordered for the compiler too, not only
            // by the wave's lockstep (the same pair as in gram_mfma_f32_kernel)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            n_sel++;
        };
        bool hit[J];
        uint32_t holders = 0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            hit[j] = m0[j] == wm; // (-0 and +0 compare equal, in the reference too)
            holders += static_cast<uint32_t>(__builtin_popcountll(__ballot(hit[j])));
        }
        if (holders == 1) {
            take(hit);
            continue;
        }
        // a tie: the lowest position in `remaining` wins
        if (lane == 0) {
            for (uint32_t i = replayed; i < n_sel; ++i) { // remaining.swap_remove(position of pick i)
                const uint32_t w = s_log[i].x, n = P - i;
                const uint32_t p = s_posof[w], moved = s_rem[n - 1];
                s_rem[p] = static_cast<uint16_t>(moved);
                s_posof[moved] = static_cast<uint16_t>(p);
            }
        }
        // lane 0's replay above is read by all 64 lanes below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        replayed = n_sel;
        uint32_t pj[J], bp = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            pj[j] = hit[j] ? static_cast<uint32_t>(s_posof[idx[j]]) : 0xFFFFFFFFu;
            bp = min(bp, pj[j]);
        }
        const uint32_t wp = wave_min_u32(bp);
        bool first[J];
#pragma unroll
        for (int j = 0; j < J; ++j)
            first[j] = pj[j] == wp;
        take(first);
    }
    for (uint32_t i = lane; i < n_sel; i += 64) {
        const uint2 e = s_log[i];
        out_order[i] = e.x;
        out_mmr[i] = __builtin_bit_cast(float, e.y);
    }
    if (lane == 0)
        *out_n = n_sel;
    if (emit.h_out) // the picks straight into the caller's (pinned) result block
        emit_result_block(emit, n_sel, lane, [&](uint32_t i) { return s_log[i].x; });
}

uint32_t ew_blocks(size_t total)
{
    size_t b = (total + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    return b ? static_cast<uint32_t>(b) : 1u;
}

} // namespace

hipError_t launch_rescore(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, const float *query,
                          const uint32_t *cand, const SelectState *st, uint64_t *packed_out, uint32_t n_pad,
                          hipStream_t s)
{
    const uint32_t blocks = (n_pad + 63) / 64;
    const size_t lds = static_cast<size_t>(dim) * sizeof(float);
    const float4 *r4 = static_cast<const float4 *>(rows);
    if (dtype == RLR_F16)
        hipLaunchKernelGGL(rescore_kernel<true>, dim3(blocks), dim3(64), lds, s, r4, pitch16, dim, query, cand, st,
                           packed_out, n_pad);
    else
        hipLaunchKernelGGL(rescore_kernel<false>, dim3(blocks), dim3(64), lds, s, r4, pitch16, dim, query, cand, st,
                           packed_out, n_pad);
    return hipGetLastError();
}

// returns false when the staged layout does not fit LDS for this row size (caller uses launch_rescore)
bool launch_rescore_staged(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, const float *query,
                           const uint32_t *cand, const SelectState *st, uint64_t *packed_out, uint32_t n_max,
                           uint32_t *hist_clear, hipStream_t s, hipError_t *err)
{
    const size_t q_bytes = static_cast<size_t>((dim + 7) & ~7u) * sizeof(float);
    const size_t row_bytes = q_bytes + 16; // f32 products of one candidate row (+16 B pad)
    if (pitch16 * (dtype == RLR_F16 ? 8u : 4u) < ((dim + 7) & ~7u))
        return false; // row pitch narrower than the 8-float rounding of dim (f32 rows, dim % 8 in 1..4)
    uint32_t cpb = 8; // 8 rows x 3 KB per workgroup: ~13 workgroups share a top-100 re-score
    while (cpb > 1 && q_bytes + cpb * row_bytes > 60 * 1024)
        cpb >>= 1;
    if (q_bytes + cpb * row_bytes > 60 * 1024)
        return false;
    const uint32_t blocks = std::max<uint32_t>((n_max + cpb - 1) / cpb, (2 * kHistBins + 255) / 256);
    const size_t lds = q_bytes + cpb * row_bytes;
    const float4 *r4 = static_cast<const float4 *>(rows);
    if (dtype == RLR_F16)
        hipLaunchKernelGGL(rescore_staged_kernel<true>, dim3(blocks), dim3(256), lds, s, r4, pitch16, dim, query, cand,
                           st, packed_out, cpb, hist_clear);
    else
        hipLaunchKernelGGL(rescore_staged_kernel<false>, dim3(blocks), dim3(256), lds, s, r4, pitch16, dim, query, cand,
                           st, packed_out, cpb, hist_clear);
    *err = hipGetLastError();
    return true;
}

bool batch_rescore_fits(uint32_t pitch16, uint32_t dim, int dtype)
{
    const size_t q_bytes = static_cast<size_t>((dim + 7) & ~7u) * sizeof(float);
    if (pitch16 * (dtype == RLR_F16 ? 8u : 4u) < ((dim + 7) & ~7u))
        return false;
    return 2 * q_bytes + 16 <= 60 * 1024; // the query + one candidate row of products
}

// (Round 3 tried two "wide" batched re-scores -- one lane per candidate over 128-byte row pieces of 256 candidates, and over the whole
// staged rows of 32 candidates -- to get rid of the one-adding-lane-per-candidate shape below.  Both were bit-identical and both SLOWER
// on the same box: config 3's finish 0.23 instead of 0.14 ms per 256 queries, config 5's 1.39 instead of 1.06 ms per 1024: the
// piecewise form pays a DRAM page activation per 128 bytes, the whole-row form serialises 32 LDS-latency-bound chains per workgroup.
// The staged kernel stays.)
// batched staged re-score; false when the staged layout does not fit LDS for this row size
bool launch_batch_rescore(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, const float *queries,
                          uint32_t q_pitch, uint32_t n_queries, uint64_t *band, uint32_t band_stride,
                          const SelectState *st, hipStream_t s, hipError_t *err)
{
    const size_t q_bytes = static_cast<size_t>((dim + 7) & ~7u) * sizeof(float);
    const size_t row_bytes = q_bytes + 16;
    if (pitch16 * (dtype == RLR_F16 ? 8u : 4u) < ((dim + 7) & ~7u))
        return false;
    uint32_t cpb = 8;
    while (cpb > 1 && q_bytes + cpb * row_bytes > 60 * 1024)
        cpb >>= 1;
    if (q_bytes + cpb * row_bytes > 60 * 1024)
        return false;
    const size_t lds = q_bytes + cpb * row_bytes;
    const float4 *r4 = static_cast<const float4 *>(rows);
    const dim3 grid(32, n_queries); // 32 groups of cpb candidates cover the usual band (~130 rows) in one sweep
    if (dtype == RLR_F16)
        hipLaunchKernelGGL(batch_rescore_kernel<true>, grid, dim3(256), lds, s, r4, pitch16, dim, queries, q_pitch, band,
                           band_stride, st, cpb);
    else
        hipLaunchKernelGGL(batch_rescore_kernel<false>, grid, dim3(256), lds, s, r4, pitch16, dim, queries, q_pitch, band,
                           band_stride, st, cpb);
    *err = hipGetLastError();
    return true;
}

hipError_t launch_score_rows(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, const float *query,
                             const uint32_t *list, uint32_t n, float *cos_out, hipStream_t s, const uint32_t *n_dev,
                             uint32_t n_rows_clamp)
{
    if (n == 0)
        return hipSuccess;
    const float4 *r4 = static_cast<const float4 *>(rows);
    // staged form when it fits LDS (the conditions of launch_rescore_staged)
    const size_t q_bytes = static_cast<size_t>((dim + 7) & ~7u) * sizeof(float);
    const size_t row_bytes = q_bytes + 16;
    uint32_t cpb = 8;
    while (cpb > 1 && q_bytes + cpb * row_bytes > 60 * 1024)
        cpb >>= 1;
    const bool fits = q_bytes + cpb * row_bytes <= 60 * 1024 && pitch16 * (dtype == RLR_F16 ? 8u : 4u) >= ((dim + 7) & ~7u);
    if (fits && n > 64) {
        const uint32_t blocks = (n + cpb - 1) / cpb;
        const size_t lds = q_bytes + cpb * row_bytes;
        if (dtype == RLR_F16)
            hipLaunchKernelGGL(score_rows_staged_kernel<true>, dim3(blocks), dim3(256), lds, s, r4, pitch16, dim, query, list, n,
                               cos_out, cpb, n_dev, n_rows_clamp);
        else
            hipLaunchKernelGGL(score_rows_staged_kernel<false>, dim3(blocks), dim3(256), lds, s, r4, pitch16, dim, query, list, n,
                               cos_out, cpb, n_dev, n_rows_clamp);
        return hipGetLastError();
    }
    const uint32_t blocks = (n + 63) / 64;
    const size_t lds = static_cast<size_t>(dim) * sizeof(float);
    if (dtype == RLR_F16)
        hipLaunchKernelGGL(score_rows_kernel<true>, dim3(blocks), dim3(64), lds, s, r4, pitch16, dim, query, list, n,
                           cos_out, n_dev, n_rows_clamp);
    else
        hipLaunchKernelGGL(score_rows_kernel<false>, dim3(blocks), dim3(64), lds, s, r4, pitch16, dim, query, list, n,
                           cos_out, n_dev, n_rows_clamp);
    return hipGetLastError();
}

hipError_t launch_normalize_store(float *staging, uint32_t n, uint32_t dim, int do_normalize, void *rows_out,
                                  uint32_t pitch16, int dtype, float *norm_tmp, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    // always: with do_normalize == 0 the caller still reads the row norms back (the guard band of unnormalised corpora)
    hipLaunchKernelGGL(sumsq_kernel, dim3((n + 63) / 64), dim3(64), 0, s, staging, n, dim, norm_tmp);
    const size_t total = static_cast<size_t>(n) * pitch16 * (dtype == RLR_F16 ? 8 : 4);
    if (dtype == RLR_F16)
        hipLaunchKernelGGL(scale_store_kernel<true>, dim3(ew_blocks(total)), dim3(256), 0, s, staging, norm_tmp, n,
                           dim, do_normalize, rows_out, pitch16);
    else
        hipLaunchKernelGGL(scale_store_kernel<false>, dim3(ew_blocks(total)), dim3(256), 0, s, staging, norm_tmp, n,
                           dim, do_normalize, rows_out, pitch16);
    return hipGetLastError();
}

hipError_t launch_synth(void *rows_out, uint32_t pitch16, uint32_t dim, int dtype, uint64_t row0, uint32_t n,
                        uint64_t seed, uint32_t n_clusters, float *norm_tmp, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    // same seed pre-hash as the oracle
    uint64_t z = seed ^ 0x5EED5EED5EED5EEDULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    hipLaunchKernelGGL(synth_sumsq_kernel, dim3((n + 63) / 64), dim3(64), 0, s, z, row0, n, dim, n_clusters,
                       norm_tmp);
    const size_t total = static_cast<size_t>(n) * pitch16 * (dtype == RLR_F16 ? 8 : 4);
    if (dtype == RLR_F16)
        hipLaunchKernelGGL(synth_store_kernel<true>, dim3(ew_blocks(total)), dim3(256), 0, s, z, row0, n, dim,
                           n_clusters, norm_tmp, rows_out, pitch16);
    else
        hipLaunchKernelGGL(synth_store_kernel<false>, dim3(ew_blocks(total)), dim3(256), 0, s, z, row0, n, dim,
                           n_clusters, norm_tmp, rows_out, pitch16);
    return hipGetLastError();
}

hipError_t launch_gather_f32(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, const uint32_t *list,
                             uint32_t n, float *out, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    const size_t total = static_cast<size_t>(n) * dim;
    if (dtype == RLR_F16)
        hipLaunchKernelGGL(gather_f32_kernel<true>, dim3(ew_blocks(total)), dim3(256), 0, s, rows, pitch16, dim, list,
                           n, out);
    else
        hipLaunchKernelGGL(gather_f32_kernel<false>, dim3(ew_blocks(total)), dim3(256), 0, s, rows, pitch16, dim,
                           list, n, out);
    return hipGetLastError();
}

hipError_t launch_compact_rows(const void *src, void *dst, uint32_t pitch16, const uint32_t *keep, uint32_t n_keep,
                               hipStream_t s)
{
    if (n_keep == 0)
        return hipSuccess;
    const size_t total = static_cast<size_t>(n_keep) * pitch16;
    hipLaunchKernelGGL(compact_rows_kernel, dim3(ew_blocks(total)), dim3(256), 0, s,
                       static_cast<const float4 *>(src), static_cast<float4 *>(dst), pitch16, keep, n_keep);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// The same reference-order Gram matrices on the f32 MATRIX cores, for binary16 rows.  A K = 1 f32 matrix instruction
// computes d = fl(c + a * b) per element, one correctly rounded step; the product of two binary16 values is exact in f32
// (11-bit x 11-bit significands), so fl(c + a * b) IS the reference's `s = s + fl(a * b)` and a chain of these instructions
// over k = 0 .. dim-1 is dot_product's strict left-to-right sum, bit for bit -- for 2 x 32 x 32 pairs per instruction
// (v_mfma_f32_32x32x1_2b_f32: lanes 0-31 carry block 0's row elements, lanes 32-63 block 1's).  Checked on the device
// against the sequential chain over 2 x 10^5 pairs with cancellation, tiny terms and binary16 subnormals
// (scratch/mmr_nom/mfma_f32_chain_probe.hip: 0 differ; the K = 2 and K = 4 forms step through k in order too).  The f32
// matrix rate equals the f32 vector rate, but an instruction needs two operand registers for 2048 products where the
// register-tiled VALU kernel above is bound by its LDS fragment reads (30 of 157 TFLOP/s): 3.3 -> ~1 ms per 1024 pools of
// 308 x 1024-d (config 5).  f32-stored rows stay on the VALU kernel: their products are not exact, and a fused step would
// round once where the reference rounds twice.
// One wave = two 32 x 32 tiles (bi <= bj) of one pool; a lane streams its A row and its B row 16 bytes (8 k-steps) at a
// time, next piece in flight behind the current 8 instructions; all tiles of a pool run on one XCD (its rows stay in that L2).
// ---------------------------------------------------------------------------------------------------------------------
typedef float v32f __attribute__((ext_vector_type(32)));

// Workgroup = ONE wave = the two tiles (2p, J) and (2p + 1, J), J >= 2p: lanes 0-31 carry the first, lanes 32-63 the second.
// Its 64 + 32 rows are staged through a wave-private LDS block per K chunk with coalesced loads (consecutive lanes cover a row's
// piece); the next chunk waits in registers behind the current chunk's matrix instructions.  No barrier anywhere: with
// four-wave workgroups sharing their staged rows the waves of the two resident workgroups ran in lock step from barrier to
// barrier and ragged workgroups left SIMDs idle (57 % matrix-pipe occupancy, 1.58 ms per 1024 pools of 300 x 1024-d); and a lane
// streaming its own row straight from global memory makes 64 cache-line requests per instruction (4.3 ms, worse than the VALU
// kernel's 3.3).  The row pair's one tile below the diagonal is computed twice, everything else once; all waves of a pool run
// on one XCD (its rows come from HBM once).
// UPR = 16-byte units per row and chunk: the K chunk is 8 UPR binary16 elements (UPR = 4: 7.7 KB of LDS per wave, five waves per SIMD
// fit; UPR = 8: 13.8 KB, 2.75 per SIMD -- RLR_GRAM_CHUNK=64 selects it for A/B runs).
template <int UPR>
__global__ __launch_bounds__(64) void gram_mfma_f32_kernel(const unsigned char *__restrict__ rows, uint32_t pitch_bytes,
                                                           const uint32_t *__restrict__ list, uint32_t P, uint32_t n_pools,
                                                           uint32_t jobs_per_pool, float *__restrict__ gram)
{
    constexpr int kGmPitch = UPR * 16 + 16; // bytes per staged row: 16-byte reads of consecutive rows land in different slots
    constexpr int kPer = 96 * UPR / 64;     // 16-byte units per lane and chunk
    __shared__ __attribute__((aligned(16))) unsigned char s_rows[96 * kGmPitch]; // rows 0..63: A (2p, 2p + 1), 64..95: B (J)
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t pool = (slot / jobs_per_pool) * 8 + xcd;
    if (pool >= n_pools)
        return;
    const uint32_t nb = (P + 31) / 32;
    uint32_t p = 0, rem = slot % jobs_per_pool; // job -> (row pair p, column block 2p + rem): pair p owns nb - 2p jobs
    while (rem >= nb - 2 * p) {
        rem -= nb - 2 * p;
        ++p;
    }
    const uint32_t I0 = 2 * p, J = 2 * p + rem;
    list += static_cast<size_t>(pool) * P;
    gram += static_cast<size_t>(pool) * P * P;
    const uint32_t lane = threadIdx.x;
    // kPer 16-byte units per lane and chunk: unit u = staged row u / UPR (0..63: A, 64..95: B), segment u % UPR
    const unsigned char *src[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const uint32_t u = lane + 64 * i, rr = u / UPR, seg = u % UPR; // rr = staged row 0..95
        const uint32_t row = min(rr < 64 ? I0 * 32 + rr : J * 32 + (rr - 64), P - 1);
        src[i] = rows + static_cast<size_t>(list[row]) * pitch_bytes + seg * 16;
    }
    const uint32_t seg = lane % UPR;
    const uint32_t row_units = pitch_bytes / 16; // a row's last chunk may be short: units beyond the pitch read as zero
    const uint32_t n_chunks = (row_units + UPR - 1) / UPR;
    uint4 pre[kPer];
    auto fetch = [&](uint32_t c) {
        const bool in = c * UPR + seg < row_units;
#pragma unroll
        for (int i = 0; i < kPer; ++i)
            pre[i] = in ? *reinterpret_cast<const uint4 *>(src[i] + static_cast<size_t>(c) * (UPR * 16)) : make_uint4(0u, 0u, 0u, 0u);
    };
    v32f acc;
#pragma unroll
    for (int i = 0; i < 32; ++i)
        acc[i] = 0.0f;
    const unsigned char *la = s_rows + lane * kGmPitch;               // A row = lane (block = lane >> 5)
    const unsigned char *lb = s_rows + (64 + (lane & 31)) * kGmPitch; // B row
    fetch(0);
    for (uint32_t c = 0; c < n_chunks; ++c) {
#pragma unroll
        for (int i = 0; i < kPer; ++i)
            *reinterpret_cast<uint4 *>(s_rows + ((lane + 64 * i) / UPR) * kGmPitch + seg * 16) = pre[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the block is private to this wave: LDS is in order, no barrier
        if (c + 1 < n_chunks)
            fetch(c + 1);
#pragma unroll
        for (int u = 0; u < UPR; ++u) {
            const uint4 ca = *reinterpret_cast<const uint4 *>(la + u * 16), cb = *reinterpret_cast<const uint4 *>(lb + u * 16);
            const uint32_t wa[4] = {ca.x, ca.y, ca.z, ca.w}, wb[4] = {cb.x, cb.y, cb.z, cb.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x1f32(h2f(static_cast<uint16_t>(wa[j] & 0xFFFF)), h2f(static_cast<uint16_t>(wb[j] & 0xFFFF)), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x1f32(h2f(static_cast<uint16_t>(wa[j] >> 16)), h2f(static_cast<uint16_t>(wb[j] >> 16)), acc, 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // this chunk's reads before the next chunk's writes
    }
    // accumulator register v of lane l: block v / 16, A row 8 ((v % 16) / 4) + 4 (l / 32) + v % 4, B row l % 32
    const uint32_t jcol = J * 32 + (lane & 31);
#pragma unroll
    for (int v = 0; v < 32; ++v) {
        const uint32_t i = (I0 + v / 16) * 32 + 8 * ((v % 16) / 4) + 4 * (lane >> 5) + (v % 4);
        if (i < P && jcol < P) {
            const float g = gram_entry(acc[v]);
            gram[static_cast<size_t>(i) * P + jcol] = g;
            gram[static_cast<size_t>(jcol) * P + i] = g;
        }
    }
}

template <int SRC>
static hipError_t launch_gram_src(const float *pool, uint32_t P, uint32_t dim, float *gram, uint32_t n_queries,
                                  const void *rows, uint32_t pitch16, const uint32_t *list, hipStream_t s)
{
    if (n_queries >= 8) {
        const uint32_t nb = (P + 63) / 64;
        hipLaunchKernelGGL((gram_tiled_kernel<4, SRC, 64>), dim3(nb * (nb + 1) / 2, 1, n_queries), dim3(256), 0, s, pool, P, dim,
                           gram, rows, pitch16, list);
    } else {
        // a single pool (or a few): the tile size that keeps the most CUs busy.  A 300-row pool is 55 blocks of 32 x 32
        // pairs -- 55 of 256 CUs, one wave per SIMD, every k-step waiting for its LDS fragments -- or 210 blocks of
        // 16 x 16 (one pair per thread): 31 -> 17 us at 768-d.  The small tile wins until its blocks outnumber the
        // CUs about six times (measured: 800-row pools still, 1024-row pools no longer; 7 pools of 300 still).
        const uint32_t nb1 = (P + 15) / 16;
        static const bool short_chunks = getenv("RLR_GRAM_SHORT_CHUNKS") != nullptr;
        if (static_cast<uint64_t>(nb1) * (nb1 + 1) / 2 * n_queries <= 512 && !short_chunks) {
            // at most two blocks per CU: 384 columns per chunk, two rounds of loads per 768-d pool instead of six (10.3 against
            // 11.9 us; before the staged loads were batched -- gram_load4 -- the same pool took 17.9)
            hipLaunchKernelGGL((gram_tiled_kernel<1, SRC, 384>), dim3(nb1 * (nb1 + 1) / 2, 1, n_queries), dim3(256), 0, s, pool, P,
                               dim, gram, rows, pitch16, list);
        } else if (static_cast<uint64_t>(nb1) * (nb1 + 1) / 2 * n_queries <= 1536) {
            hipLaunchKernelGGL((gram_tiled_kernel<1, SRC, 128>), dim3(nb1 * (nb1 + 1) / 2, 1, n_queries), dim3(256), 0, s, pool, P,
                               dim, gram, rows, pitch16, list);
        } else {
            const uint32_t nb = (P + 31) / 32;
            hipLaunchKernelGGL((gram_tiled_kernel<2, SRC, 64>), dim3(nb * (nb + 1) / 2, 1, n_queries), dim3(256), 0, s, pool, P, dim,
                               gram, rows, pitch16, list);
        }
    }
    return hipGetLastError();
}

hipError_t launch_gram(const float *pool, uint32_t P, uint32_t dim, float *gram, uint32_t n_queries, hipStream_t s)
{
    if (P == 0 || n_queries == 0)
        return hipSuccess;
    static const bool naive = getenv("RLR_GRAM_NAIVE") != nullptr;
    if (naive) {
        hipLaunchKernelGGL(gram_kernel, dim3((P + 63) / 64, P, n_queries), dim3(64), 0, s, pool, P, dim, gram);
        return hipGetLastError();
    }
    return launch_gram_src<0>(pool, P, dim, gram, n_queries, nullptr, 0, nullptr, s);
}

// the same Gram matrices with the pool rows read from the index through `list` (n_queries x P row numbers)
hipError_t launch_gram_rows(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, const uint32_t *list, uint32_t P,
                            float *gram, uint32_t n_queries, hipStream_t s)
{
    if (P == 0 || n_queries == 0)
        return hipSuccess;
    if (dtype == RLR_F16) {
        // enough pools to fill the chip (two 32 x 32 tiles per wave, 1024 k-steps each): the f32 matrix cores
        static const bool valu_only = getenv("RLR_GRAM_VALU") != nullptr;
        if (n_queries >= 16 && !valu_only) {
            const uint32_t nb = (P + 31) / 32;
            uint32_t jobs_per_pool = 0;
            for (uint32_t pr = 0; 2 * pr < nb; ++pr)
                jobs_per_pool += nb - 2 * pr;
            const uint32_t pools8 = (n_queries + 7) / 8 * 8;
            static const bool wide_chunk = getenv("RLR_GRAM_CHUNK") && atoi(getenv("RLR_GRAM_CHUNK")) == 64;
            if (wide_chunk)
                hipLaunchKernelGGL(gram_mfma_f32_kernel<8>, dim3(pools8 * jobs_per_pool), dim3(64), 0, s, static_cast<const unsigned char *>(rows),
                                   pitch16 * 16u, list, P, n_queries, jobs_per_pool, gram);
            else
                hipLaunchKernelGGL(gram_mfma_f32_kernel<4>, dim3(pools8 * jobs_per_pool), dim3(64), 0, s, static_cast<const unsigned char *>(rows),
                                   pitch16 * 16u, list, P, n_queries, jobs_per_pool, gram);
            return hipGetLastError();
        }
        return launch_gram_src<2>(nullptr, P, dim, gram, n_queries, rows, pitch16, list, s);
    }
    return launch_gram_src<1>(nullptr, P, dim, gram, n_queries, rows, pitch16, list, s);
}

// n_queries > 1 (or sizes != null): per-query arrays strided by P, pool sizes in sizes[q] (<= P <= 1024)
hipError_t launch_mmr_greedy(const float *gram, const float *scores, uint32_t P, uint32_t k, float lambda,
                             uint32_t *out_order, float *out_mmr, uint32_t *out_n, const uint32_t *sizes,
                             uint32_t n_queries, hipStream_t s, const MmrEmit *emit_in)
{
    const size_t lds = static_cast<size_t>(P ? P : 1) * 12;
    const MmrEmit emit = emit_in ? *emit_in : MmrEmit{};
    if (emit.h_out && (n_queries != 1 || P == 0 || P > 1024))
        return hipErrorInvalidValue; // the emit tail exists in the register-resident single-pool kernel only
    // RLR_MMR_GREEDY=reg: every candidate's position kept in registers (the kernel before the lazy tie-break)
    static const char *g_env = getenv("RLR_MMR_GREEDY");
    static const bool reg = g_env && !strcmp(g_env, "reg");
#define RLR_MMR_REG(JV)                                                                                          \
    do {                                                                                                         \
        if (reg)                                                                                                 \
            hipLaunchKernelGGL(mmr_greedy_reg_kernel<JV>, dim3(n_queries), dim3(256), 0, s, gram, scores, P, k, lambda, \
                               out_order, out_mmr, out_n, sizes, emit);                                          \
        else                                                                                                     \
            hipLaunchKernelGGL(mmr_greedy_lazy_kernel<JV>, dim3(n_queries), dim3(256), 0, s, gram, scores, P, k, lambda, \
                               out_order, out_mmr, out_n, sizes, emit);                                          \
    } while (0)
    if (P == 0 || (P > 1024 && n_queries == 1 && !sizes))
        hipLaunchKernelGGL(mmr_greedy_kernel, dim3(1), dim3(64), lds, s, gram, scores, P, k, lambda, out_order, out_mmr,
                           out_n);
    else if (P > 1024)
        return hipErrorInvalidValue;
    else if (P <= 64)
        RLR_MMR_REG(1);
    else if (P <= 128)
        RLR_MMR_REG(2);
    else if (P <= 320)
        RLR_MMR_REG(5);
    else if (P <= 512)
        RLR_MMR_REG(8);
    else
        RLR_MMR_REG(16);
#undef RLR_MMR_REG
    return hipGetLastError();
}

} // namespace rlr
