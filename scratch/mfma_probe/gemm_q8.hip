// gemm_q8.hip -- batched nomination on the int8 matrix cores (optional, on top of the 8-bit copy of q8.hip).
//
// The batched GEMM of gemm.hip only nominates too, so it can run on v_mfma_i32_16x16x64_i8: the rows' 8-bit codes
// (k_r, scale s_r) against queries quantised the same way on the host (k_q, scale t_q); the integer accumulation
// is exact, the nominated score is s_r * t_q * acc, and
//     | q.x - (t_q k_q).(s_r k_r) |  <=  ||q|| * delta_r  +  ||q - t_q k_q|| * ||s_r k_r||
// with delta_r <= delta_max (kept by q8_build_kernel) and ||s_r k_r|| <= xnorm_max + delta_max.  Half the bytes of
// the binary16 image, twice the K per MFMA instruction at the same cycle count.
//
// Image layout: [tile of 256 rows][K-chunk of 64][wave 8][row group 2][lane 64][16 bytes], a fragment (16 rows x
// 64 k) is one lane-linear 1 KiB load: lane l holds row (l & 15), k = 16 * (l >> 4) + j.  Queries use the same k map:
// [qblock 256][chunk][colblock 16][lane 64][16 bytes] with query = colblock * 16 + (l & 15).
// Kernel structure = gemm_image_kernel (gemm.hip): 8 waves, wave tile 64 rows x 128 queries, query chunk staged
// through a double-buffered LDS image, A fragments in a register ring straight from the image.
#include "common.h"
#include "kernels.h"
#include "../../include/rlr_gpu.h"

#include <algorithm>

namespace rlr {

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kQ8QB = 256;             // queries per workgroup tile
constexpr int kQ8NB = kQ8QB / 16;      // 16 column blocks
constexpr int kQ8BM = 256;             // rows per workgroup tile
constexpr int kQ8RG = 4;               // row groups per wave (64 rows)
constexpr int kQ8NBW = kQ8NB / 2;      // column blocks per wave (128 queries)
constexpr int kQ8ChunkFrags = kQ8NB * 64; // i32x4 entries of one B chunk (16 KB)
constexpr int kQ8Slots = 4;            // A ring: chunks in flight per wave

struct Q8GemmArgs {
    const i32x4 *image;          // row codes, fragment-major
    const float *row_scale;      // s_r per row (NaN for rows holding a NaN)
    const i32x4 *qfrag;          // query codes, fragment-major
    const float *q_scale;        // t_q per query
    uint32_t row_begin, row_end; // multiples of 256 except the very end
    uint32_t n_chunks;           // dim / 64
    uint32_t n_qblocks, n_queries;
    const float *tau;
    uint64_t *cand;
    uint32_t cand_stride;
    SelectState *st;
    float *scores;
    size_t score_stride;
};

__global__ __launch_bounds__(256) void q8_image_kernel(const uint8_t *__restrict__ q8, uint32_t dim, uint32_t n_rows,
                                                       uint32_t n_chunks, uint32_t tile_begin, uint32_t tile_end,
                                                       i32x4 *__restrict__ image)
{
    const size_t first = static_cast<size_t>(tile_begin) * n_chunks * 8 * 2 * 64;
    const size_t total = static_cast<size_t>(tile_end - tile_begin) * n_chunks * 8 * 2 * 64;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const size_t e = first + i;
        const uint32_t lane = e & 63;
        const uint32_t rg = (e >> 6) & 1;
        const uint32_t w = (e >> 7) & 7;
        const size_t tc = e >> 10; // tile * n_chunks + chunk
        const uint32_t c = static_cast<uint32_t>(tc % n_chunks);
        const uint32_t t = static_cast<uint32_t>(tc / n_chunks);
        const uint32_t row = t * 256 + w * 32 + rg * 16 + (lane & 15);
        const uint32_t k = c * 64 + (lane >> 4) * 16;
        i32x4 v = {0, 0, 0, 0}; // code 0 for padding rows
        if (row < n_rows) {
            const i32x4 raw = *reinterpret_cast<const i32x4 *>(q8 + static_cast<size_t>(row) * dim + k);
            // the row-major copy stores k + 128: flipping the top bit of every byte gives the signed code
            v = i32x4{raw[0] ^ static_cast<int>(0x80808080u), raw[1] ^ static_cast<int>(0x80808080u),
                      raw[2] ^ static_cast<int>(0x80808080u), raw[3] ^ static_cast<int>(0x80808080u)};
        }
        image[e] = v;
    }
}

template <bool MATERIALISE>
__device__ __forceinline__ void q8_epilogue(const Q8GemmArgs &a, i32x4 (&acc)[kQ8RG][kQ8NBW], uint32_t row0,
                                            uint32_t last_row, uint32_t q0, int lane)
{
    const uint32_t qcol = q0 + (lane & 15);
    float tq[kQ8NBW], tau_l[kQ8NBW];
#pragma unroll
    for (int nb = 0; nb < kQ8NBW; ++nb) {
        const uint32_t q = qcol + nb * 16;
        tq[nb] = q < a.n_queries ? a.q_scale[q] : 0.0f;
        tau_l[nb] = (!MATERIALISE && q < a.n_queries) ? a.tau[q] : __builtin_inff();
    }
#pragma unroll
    for (int rg = 0; rg < kQ8RG; ++rg) {
        const uint32_t r = row0 + rg * 16 + 4 * (lane >> 4);
        float sr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            sr[i] = r + i <= last_row ? a.row_scale[r + i] : 0.0f;
#pragma unroll
        for (int nb = 0; nb < kQ8NBW; ++nb) {
            const uint32_t q = qcol + nb * 16;
            const i32x4 v = acc[rg][nb];
            float f[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                f[i] = sr[i] * (tq[nb] * static_cast<float>(v[i]));
            if constexpr (MATERIALISE) {
                if (q >= a.n_queries)
                    continue;
                float *dst = a.scores + static_cast<size_t>(q) * a.score_stride + (r - a.row_begin);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (r + i <= last_row)
                        dst[i] = f[i];
            } else {
                const float t = tau_l[nb];
                if (f[0] >= t || f[1] >= t || f[2] >= t || f[3] >= t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (f[i] >= t && r + i <= last_row) {
                            const uint32_t slot = atomicAdd(&a.st[q].n_cand, 1u);
                            if (slot < a.st[q].cap)
                                a.cand[static_cast<size_t>(q) * a.cand_stride + slot] = pack_result(f[i], r + i);
                        }
                    }
                }
            }
        }
    }
}

template <bool MATERIALISE>
__global__ __launch_bounds__(512) void gemm_q8_kernel(const Q8GemmArgs a)
{
    __shared__ i32x4 s_b[2][kQ8ChunkFrags];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wq = wave & 1;
    const uint32_t bid = blockIdx.x;
    const uint32_t rt = (bid / (8 * a.n_qblocks)) * 8 + (bid & 7);
    const uint32_t qb = (bid >> 3) % a.n_qblocks;
    const uint32_t n_rows = a.row_end - a.row_begin;
    if (rt * kQ8BM >= n_rows)
        return;
    const uint32_t row0 = a.row_begin + rt * kQ8BM + wr * (kQ8RG * 16);
    const uint32_t last_row = a.row_end - 1;
    const uint32_t n_chunks = a.n_chunks;
    const uint32_t tile = a.row_begin / kQ8BM + rt;
    // rows 64*wr + 16*rg + r of the tile are image wave (2*wr + rg/2), row group rg % 2
    const i32x4 *ap = a.image + (static_cast<size_t>(tile) * n_chunks * 8 + 2 * wr) * 2 * 64 + lane;
    const i32x4 *bsrc = a.qfrag + static_cast<size_t>(qb) * n_chunks * kQ8ChunkFrags;

    i32x4 acc[kQ8RG][kQ8NBW];
#pragma unroll
    for (int rg = 0; rg < kQ8RG; ++rg)
#pragma unroll
        for (int nb = 0; nb < kQ8NBW; ++nb)
            acc[rg][nb] = i32x4{0, 0, 0, 0};

    i32x4 breg[2];
    i32x4 ring[kQ8Slots][kQ8RG];
    const uint32_t last_c = n_chunks - 1;
    const uint32_t c_rot = (bid * 5u) % n_chunks; // per-workgroup K rotation (see gemm_image_kernel)
#define Q8_ROT(X) (((X) + c_rot) % n_chunks)
#define Q8_LOAD_B(CHUNK) \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) breg[i] = bsrc[static_cast<size_t>(CHUNK) * kQ8ChunkFrags + tid + 512 * i]
#define Q8_STORE_B(BUF) _Pragma("unroll") for (int i = 0; i < 2; ++i) s_b[BUF][tid + 512 * i] = breg[i]
#define Q8_LOAD_A(CHUNK, SLOT) \
    _Pragma("unroll") for (int rg = 0; rg < kQ8RG; ++rg) ring[SLOT][rg] = ap[static_cast<size_t>(CHUNK) * (8 * 2 * 64) + rg * 64]

#pragma unroll
    for (int sl = 0; sl < kQ8Slots; ++sl) {
        Q8_LOAD_A(Q8_ROT(min(static_cast<uint32_t>(sl), last_c)), sl);
        __builtin_amdgcn_sched_barrier(0); // issue order = consumption order
    }
    Q8_LOAD_B(Q8_ROT(0));
    Q8_STORE_B(0);
    __syncthreads();

    // n_chunks is a multiple of kQ8Slots: branch-free, slot and LDS buffer indices are compile-time constants
#pragma unroll 1
    for (uint32_t c = 0; c < n_chunks; c += kQ8Slots) {
#pragma unroll
        for (int sl = 0; sl < kQ8Slots; ++sl) {
            const uint32_t cc = c + sl;
            Q8_LOAD_B(Q8_ROT(min(cc + 1, last_c)));
            __builtin_amdgcn_sched_barrier(0);
            const i32x4 *sb = s_b[sl & 1] + wq * (kQ8NBW * 64) + lane;
#pragma unroll
            for (int nb = 0; nb < kQ8NBW; ++nb) {
                const i32x4 b = sb[nb * 64];
#pragma unroll
                for (int rg = 0; rg < kQ8RG; ++rg)
                    acc[rg][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[sl][rg], b, acc[rg][nb], 0, 0, 0);
            }
            Q8_LOAD_A(Q8_ROT(min(cc + kQ8Slots, last_c)), sl);
            __builtin_amdgcn_sched_barrier(0); // keep the refill here and the LDS stores below it
            Q8_STORE_B((sl & 1) ^ 1);
            __syncthreads();
        }
    }
#undef Q8_LOAD_A
#undef Q8_STORE_B
#undef Q8_LOAD_B
#undef Q8_ROT
    q8_epilogue<MATERIALISE>(a, acc, row0, last_row, qb * kQ8QB + wq * (kQ8NBW * 16), lane);
}

} // namespace

size_t q8_image_bytes(uint32_t dim, uint64_t n_rows)
{
    const uint64_t tiles = (n_rows + kQ8BM - 1) / kQ8BM;
    return static_cast<size_t>(tiles) * kQ8BM * dim;
}

hipError_t launch_q8_image_build(const void *q8, uint32_t dim, uint32_t n_rows, uint32_t tile_begin, uint32_t tile_end,
                                 void *image, hipStream_t s)
{
    if (tile_end <= tile_begin)
        return hipSuccess;
    const uint32_t n_chunks = dim / 64;
    const size_t total = static_cast<size_t>(tile_end - tile_begin) * n_chunks * 8 * 2 * 64;
    const uint32_t blocks = static_cast<uint32_t>(std::min<size_t>((total + 255) / 256, 65536));
    hipLaunchKernelGGL(q8_image_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const uint8_t *>(q8), dim, n_rows, n_chunks,
                       tile_begin, tile_end, static_cast<i32x4 *>(image));
    return hipGetLastError();
}

// fragment-major query codes for launch_gemm_q8: [qblock][chunk][colblock][lane][16 bytes]; host side, the
// caller uploads the buffer.  codes: n_queries x dim signed bytes.
void q8_pack_queries(const int8_t *codes, uint32_t n_queries, uint32_t dim, int8_t *frag)
{
    const uint32_t n_chunks = dim / 64;
    const uint32_t n_qblocks = (n_queries + kQ8QB - 1) / kQ8QB;
    for (uint32_t qb = 0; qb < n_qblocks; ++qb)
        for (uint32_t c = 0; c < n_chunks; ++c)
            for (uint32_t nb = 0; nb < kQ8NB; ++nb)
                for (uint32_t lane = 0; lane < 64; ++lane) {
                    int8_t *dst = frag + ((((static_cast<size_t>(qb) * n_chunks + c) * kQ8NB + nb) * 64) + lane) * 16;
                    const uint32_t q = qb * kQ8QB + nb * 16 + (lane & 15);
                    const uint32_t k = c * 64 + (lane >> 4) * 16;
                    if (q < n_queries)
                        std::copy(codes + static_cast<size_t>(q) * dim + k, codes + static_cast<size_t>(q) * dim + k + 16, dst);
                    else
                        std::fill(dst, dst + 16, static_cast<int8_t>(0));
                }
}

size_t q8_query_frag_bytes(uint32_t n_queries, uint32_t dim)
{
    return static_cast<size_t>((n_queries + kQ8QB - 1) / kQ8QB) * kQ8QB * dim;
}

hipError_t launch_gemm_q8(const void *image, const float *row_scale, uint32_t dim, uint32_t row_begin, uint32_t row_end,
                          const void *qfrag, const float *q_scale, uint32_t n_queries, const float *tau, uint64_t *cand,
                          uint32_t cand_stride, SelectState *st, float *scores, size_t score_stride, hipStream_t s)
{
    if (row_end <= row_begin)
        return hipSuccess;
    Q8GemmArgs a;
    a.image = static_cast<const i32x4 *>(image);
    a.row_scale = row_scale;
    a.qfrag = static_cast<const i32x4 *>(qfrag);
    a.q_scale = q_scale;
    a.row_begin = row_begin;
    a.row_end = row_end;
    a.n_chunks = dim / 64;
    a.n_qblocks = (n_queries + kQ8QB - 1) / kQ8QB;
    a.n_queries = n_queries;
    a.tau = tau;
    a.cand = cand;
    a.cand_stride = cand_stride;
    a.st = st;
    a.scores = scores;
    a.score_stride = score_stride;
    const uint32_t n_rt = (row_end - row_begin + kQ8BM - 1) / kQ8BM;
    const uint32_t grid = ((n_rt + 7) / 8) * 8 * a.n_qblocks;
    if (scores)
        hipLaunchKernelGGL((gemm_q8_kernel<true>), dim3(grid), dim3(512), 0, s, a);
    else
        hipLaunchKernelGGL((gemm_q8_kernel<false>), dim3(grid), dim3(512), 0, s, a);
    return hipGetLastError();
}

} // namespace rlr
