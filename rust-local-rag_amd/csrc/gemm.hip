// gemm.hip -- batched queries: Q x Corpus^T as a dense GEMM on the gfx950 matrix cores with a
// fused top-k nomination epilogue (the additive batched entry point of SURVEY.md section 8; its
// oracle is "loop the single-query reference over the batch").
//
// Role of the MFMA product: it NOMINATES.  Operands are rounded to binary16 (rows on the fly in
// registers, queries once into a fragment-major image), accumulated in f32 by
// v_mfma_f32_16x16x32_f16.  For unit-norm operands |nominated - reference dot| <= eps_nom
// (2^-10 from the two operand roundings + accumulation terms, see nomination_eps()), so every
// row of the true top-k lies within 2*eps_nom of the k-th nominated score; those rows are then
// re-scored in strict reference order (exact_dot.h) and sorted, and the emitted rows/scores are
// bit-identical to the single-query path and to the reference loop.
//
// Shape on gfx950 (HBM-bound: each corpus row is read once per 256 queries)
//   * workgroup = 8 waves = 256 corpus rows x 256 queries; a wave owns 32 rows (2 MFMA row
//     groups) x all 256 queries: 32 accumulator tiles of 16x16 = 128 VGPRs;
//   * corpus rows (A operand) go HBM -> registers directly in MFMA fragment shape: lane
//     (row = l&15, kgroup = l>>4) reads the 8 consecutive k it owns (32 B of f32, 16 B of f16);
//     4 lanes cover one 128-B (64-B) line, no LDS round trip for the streamed operand;
//   * queries (B operand) are pre-arranged [kstep][colblock][lane][8 halfs], so a K-chunk is one
//     linear 32 KB copy into LDS (double buffered) and every fragment read is a conflict-free
//     linear ds_read_b128;
//   * workgroups sharing a row tile (Q > 256) get consecutive ids with equal id % 8, i.e. the
//     same XCD, so the tile's second read is an L2 hit (speed only, never correctness);
//   * epilogue: compare each accumulator with its query's threshold (from a materialised sample
//     of the first S rows) and append (score, row) to that query's candidate list.
#include "common.h"
#include "exact_dot.h"
#include "kernels.h"
#include "lds_select.h"
#include "../../include/rlr_gpu.h"

#include <cstdlib>

namespace rlr {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kQB = 256;            // queries per workgroup tile
constexpr int kNB = kQB / 16;       // 16-wide query column blocks
constexpr int kRG = 2;              // 16-row groups per wave
constexpr int kGemmWaves = 8;
constexpr int kBM = kGemmWaves * kRG * 16; // 256 corpus rows per workgroup tile
constexpr int kKsChunk = 2;         // MFMA k-steps (32 k each) per LDS chunk
constexpr int kChunkFrags = kKsChunk * kNB * 64; // half8 entries per B chunk (32 KB)

// queries f32 [n_queries x q_pitch] -> binary16, fragment-major:
//   [qblock][kstep][colblock][lane][8]   with  query = qblock*256 + colblock*16 + (lane & 15)
// The k a lane's element j stands for only has to agree between the A and B fragments, so it is
// chosen for the A (corpus row) loads to be contiguous per instruction:
//   f16 rows: k = kstep*32 + g*8 + j                       (one 16-B load, 4 lanes = 64 B of a row)
//   f32 rows: k = kstep*32 + g*4 + j        for j < 4      (first 16-B load : bytes [g*16, +16))
//             k = kstep*32 + 16 + g*4 + j-4 for j >= 4     (second 16-B load: bytes [64 + g*16, +16))
// with g = lane >> 4: each load instruction covers 16 rows x 64 contiguous bytes.
__global__ __launch_bounds__(256) void prep_queries_kernel(const float *__restrict__ q, uint32_t n_queries,
                                                           uint32_t q_pitch, uint32_t dim, uint32_t n_ksteps,
                                                           uint32_t n_qblocks, int f16_rows,
                                                           _Float16 *__restrict__ out)
{
    const size_t total = static_cast<size_t>(n_qblocks) * n_ksteps * kNB * 64 * 8;
    for (size_t e = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<size_t>(gridDim.x) * 256) {
        const uint32_t j = e & 7;
        const uint32_t lane = (e >> 3) & 63;
        const uint32_t nb = (e >> 9) & (kNB - 1);
        const size_t rest = e >> 13; // qblock * n_ksteps + kstep
        const uint32_t ks = static_cast<uint32_t>(rest % n_ksteps);
        const uint32_t qb = static_cast<uint32_t>(rest / n_ksteps);
        const uint32_t query = qb * kQB + nb * 16 + (lane & 15);
        const uint32_t g = lane >> 4;
        const uint32_t k = ks * 32 + (f16_rows ? g * 8 + j : (j < 4 ? g * 4 + j : 16 + g * 4 + (j - 4)));
        float v = 0.0f;
        if (query < n_queries && k < dim)
            v = q[static_cast<size_t>(query) * q_pitch + k];
        out[e] = static_cast<_Float16>(v); // round to nearest even
    }
}

__device__ inline half8 cvt8(float4 lo, float4 hi)
{
    half8 r;
    r[0] = static_cast<_Float16>(lo.x);
    r[1] = static_cast<_Float16>(lo.y);
    r[2] = static_cast<_Float16>(lo.z);
    r[3] = static_cast<_Float16>(lo.w);
    r[4] = static_cast<_Float16>(hi.x);
    r[5] = static_cast<_Float16>(hi.y);
    r[6] = static_cast<_Float16>(hi.z);
    r[7] = static_cast<_Float16>(hi.w);
    return r;
}

struct GemmArgs {
    const unsigned char *rows; // index rows
    uint32_t pitch_bytes;
    uint32_t row_begin, row_end; // [begin, end) rows this launch covers
    uint32_t n_ksteps;           // dim / 32 (even)
    const half8 *qfrag;
    uint32_t n_qblocks, n_queries;
    const float *tau;            // filter mode: per-query threshold on the nominated score
    uint64_t *cand;              // filter mode: [query][cand_stride] packed (score, row)
    uint32_t cand_stride;
    SelectState *st;             // filter mode: per-query counters / capacity
    float *scores;               // materialise mode: [query][score_stride], column = row - row_begin
    size_t score_stride;
    uint32_t *sync;              // gemm8, several query blocks: 256 zeroed words, one arrival counter per sibling group (or null)
    uint32_t sync_every;         // ... and the siblings meet before every sync_every-th unit
    uint32_t a_nt_shared;        // (experiment, RLR_GEMM8_A_NT=1) row half-tiles streamed `nt` even when sibling workgroups share them:
                                 // config 5's share fetched 1.36 x (meeting every unit) / 1.50 x (every 2nd) the image against
                                 // 1.22 x / 1.28 x without -- the siblings are not tight enough for evict-first lines; off
};

// ---- epilogue shared by the GEMM kernels: D layout is col = lane & 15 (query), row = 4*(lane >> 4) + reg
template <bool MATERIALISE>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &a, f32x4 (&acc)[kRG][kNB], uint32_t row0, uint32_t last_row,
                                              uint32_t qb, int lane)
{
    const uint32_t qcol = qb * kQB + (lane & 15);
    if constexpr (MATERIALISE) {
#pragma unroll
        for (int rg = 0; rg < kRG; ++rg) {
            const uint32_t r = row0 + rg * 16 + 4 * (lane >> 4);
            const uint32_t rel = r - a.row_begin;
#pragma unroll
            for (int nb = 0; nb < kNB; ++nb) {
                const uint32_t q = qcol + nb * 16;
                if (q >= a.n_queries)
                    continue;
                float *dst = a.scores + static_cast<size_t>(q) * a.score_stride + rel;
                const f32x4 v = acc[rg][nb];
                if (r + 3 <= last_row) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (r + i <= last_row)
                            dst[i] = v[i];
                }
            }
        }
    } else {
        float tau_l[kNB];
#pragma unroll
        for (int nb = 0; nb < kNB; ++nb) {
            const uint32_t q = qcol + nb * 16;
            tau_l[nb] = q < a.n_queries ? a.tau[q] : __builtin_inff();
        }
#pragma unroll
        for (int rg = 0; rg < kRG; ++rg) {
            const uint32_t r = row0 + rg * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int nb = 0; nb < kNB; ++nb) {
                const f32x4 v = acc[rg][nb];
                const float t = tau_l[nb];
                if (v[0] >= t || v[1] >= t || v[2] >= t || v[3] >= t) {
                    const uint32_t q = qcol + nb * 16;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (v[i] >= t && r + i <= last_row) {
                            const uint32_t slot = atomicAdd(&a.st[q].n_cand, 1u);
                            if (slot < a.st[q].cap)
                                a.cand[static_cast<size_t>(q) * a.cand_stride + slot] = pack_result(v[i], r + i);
                        }
                    }
                }
            }
        }
    }
}

template <bool F16ROWS, bool MATERIALISE>
__global__ __launch_bounds__(512) void gemm_nominate_kernel(const GemmArgs a)
{
    __shared__ half8 s_b[2][kChunkFrags];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile map: the n_qblocks workgroups of one row tile share id % 8
    const uint32_t bid = blockIdx.x;
    const uint32_t rt = (bid / (8 * a.n_qblocks)) * 8 + (bid & 7);
    const uint32_t qb = (bid >> 3) % a.n_qblocks;
    const uint32_t n_rows = a.row_end - a.row_begin;
    if (rt * kBM >= n_rows)
        return;
    const uint32_t row0 = a.row_begin + rt * kBM + wave * (kRG * 16);
    const uint32_t last_row = a.row_end - 1;

    // per-lane A pointers (fragment shape: row = l & 15, k-group = l >> 4)
    const unsigned char *ap[kRG];
#pragma unroll
    for (int rg = 0; rg < kRG; ++rg) {
        const uint32_t r = min(row0 + rg * 16 + (lane & 15), last_row);
        ap[rg] = a.rows + static_cast<size_t>(r) * a.pitch_bytes + (lane >> 4) * 16;
    }
    constexpr int kStepBytes = F16ROWS ? 64 : 128; // bytes of one row consumed per k-step

    const uint32_t n_chunks = a.n_ksteps / kKsChunk;
    const half8 *bsrc = a.qfrag + static_cast<size_t>(qb) * a.n_ksteps * kNB * 64;

    f32x4 acc[kRG][kNB];
#pragma unroll
    for (int rg = 0; rg < kRG; ++rg)
#pragma unroll
        for (int nb = 0; nb < kNB; ++nb)
            acc[rg][nb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    half8 breg[4];
    float4 araw[kRG][kKsChunk][2]; // f32 rows: two 16-B loads per fragment
    half8 a_cur[kRG][kKsChunk];

    auto load_b = [&](uint32_t c) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            breg[i] = bsrc[static_cast<size_t>(c) * kChunkFrags + tid + 512 * i];
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            s_b[buf][tid + 512 * i] = breg[i];
    };
    auto load_a = [&](uint32_t c) {
#pragma unroll
        for (int rg = 0; rg < kRG; ++rg)
#pragma unroll
            for (int ks = 0; ks < kKsChunk; ++ks) {
                const unsigned char *p = ap[rg] + static_cast<size_t>(c * kKsChunk + ks) * kStepBytes;
                if constexpr (F16ROWS) {
                    araw[rg][ks][0] = *reinterpret_cast<const float4 *>(p);
                } else {
                    araw[rg][ks][0] = *reinterpret_cast<const float4 *>(p);
                    araw[rg][ks][1] = *reinterpret_cast<const float4 *>(p + 64);
                }
            }
    };
    auto convert_a = [&]() {
#pragma unroll
        for (int rg = 0; rg < kRG; ++rg)
#pragma unroll
            for (int ks = 0; ks < kKsChunk; ++ks) {
                if constexpr (F16ROWS)
                    a_cur[rg][ks] = __builtin_bit_cast(half8, araw[rg][ks][0]);
                else
                    a_cur[rg][ks] = cvt8(araw[rg][ks][0], araw[rg][ks][1]);
            }
    };

    load_b(0);
    load_a(0);
    store_b(0);
    convert_a();
    __syncthreads();

#pragma unroll 1
    for (uint32_t c = 0; c < n_chunks; ++c) {
        const bool has_next = c + 1 < n_chunks;
        if (has_next) {
            load_b(c + 1);
            load_a(c + 1);
        }
        const half8 *sb = s_b[c & 1];
#pragma unroll
        for (int ks = 0; ks < kKsChunk; ++ks) {
#pragma unroll
            for (int nb = 0; nb < kNB; ++nb) {
                const half8 b = sb[(ks * kNB + nb) * 64 + lane];
#pragma unroll
                for (int rg = 0; rg < kRG; ++rg)
                    acc[rg][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur[rg][ks], b, acc[rg][nb], 0, 0, 0);
            }
        }
        if (has_next) {
            store_b((c + 1) & 1);
            convert_a();
        }
        __syncthreads();
    }

    gemm_epilogue<MATERIALISE>(a, acc, row0, last_row, qb, lane);
}

// ---------------------------------------------------------------------------------------------
// Nomination image: an optional binary16 copy of the corpus laid out for this GEMM,
//   [tile of 256 rows][K-chunk of 64][wave 8][row group 2][k-step 2][lane 64][8 halfs]
// i.e. every MFMA A fragment of every wave is one lane-linear 1 KiB block, a tile's K-chunk is
// 32 KB contiguous and a whole tile is dim/64 * 32 KB contiguous.  The row-major matrix makes
// this GEMM read 256-byte pieces at a 3 KB stride (12 visits per DRAM page, every lane its own
// request); the image turns the same bytes into a sequential stream of perfectly coalesced loads
// and halves them for f32 corpora.  Costs dim*2 bytes per row of HBM; the row-major matrix stays
// the master copy (exact re-score, single-query scan).
// ---------------------------------------------------------------------------------------------
template <bool F16ROWS>
__global__ __launch_bounds__(256) void build_image_kernel(const unsigned char *__restrict__ rows, uint32_t pitch_bytes,
                                                          uint32_t n_rows, uint32_t n_chunks, uint32_t tile_begin,
                                                          uint32_t tile_end, half8 *__restrict__ image)
{
    const size_t per_tile = static_cast<size_t>(n_chunks) * 8 * 4 * 64; // half8 entries per tile
    const size_t first = static_cast<size_t>(tile_begin) * per_tile;
    const size_t total = static_cast<size_t>(tile_end - tile_begin) * per_tile;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const size_t e = first + i;
        const uint32_t lane = e & 63;
        const uint32_t ks = (e >> 6) & 1;
        const uint32_t rg = (e >> 7) & 1;
        const uint32_t w = (e >> 8) & 7;
        const size_t tc = e >> 11; // tile * n_chunks + chunk
        const uint32_t c = static_cast<uint32_t>(tc % n_chunks);
        const uint32_t t = static_cast<uint32_t>(tc / n_chunks);
        const uint32_t row = t * 256 + w * 32 + rg * 16 + (lane & 15);
        const uint32_t k = c * 64 + ks * 32 + (lane >> 4) * 8;
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = static_cast<_Float16>(0.0f);
        if (row < n_rows) {
            const unsigned char *src = rows + static_cast<size_t>(row) * pitch_bytes;
            if constexpr (F16ROWS) {
                v = *reinterpret_cast<const half8 *>(src + k * 2);
            } else {
                const float4 lo = *reinterpret_cast<const float4 *>(src + k * 4);
                const float4 hi = *reinterpret_cast<const float4 *>(src + k * 4 + 16);
                v = cvt8(lo, hi);
            }
        }
        image[e] = v;
    }
}

// Wave tiles over the image are 64 rows x (16 NBW) queries: 4 row groups of 16; A fragments come straight from the
// image (lane-linear 1 KiB loads).  Used by the resident-query kernel below; the main batched kernel (gemm8_kernel)
// stages both operands through LDS instead.
constexpr int kRGI = 4;          // row groups per wave in the image kernel

template <bool MATERIALISE, int NBW>
__device__ __forceinline__ void gemm_image_epilogue(const GemmArgs &a, f32x4 (&acc)[kRGI][NBW], uint32_t row0,
                                                    uint32_t last_row, uint32_t q0, int lane)
{
    constexpr int kNBI = NBW; // column blocks of this wave tile
    const uint32_t qcol = q0 + (lane & 15);
    if constexpr (MATERIALISE) {
#pragma unroll
        for (int rg = 0; rg < kRGI; ++rg) {
            const uint32_t r = row0 + rg * 16 + 4 * (lane >> 4);
            const uint32_t rel = r - a.row_begin;
#pragma unroll
            for (int nb = 0; nb < kNBI; ++nb) {
                const uint32_t q = qcol + nb * 16;
                if (q >= a.n_queries)
                    continue;
                float *dst = a.scores + static_cast<size_t>(q) * a.score_stride + rel;
                const f32x4 v = acc[rg][nb];
                if (r + 3 <= last_row) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (r + i <= last_row)
                            dst[i] = v[i];
                }
            }
        }
    } else {
        float tau_l[kNBI];
#pragma unroll
        for (int nb = 0; nb < kNBI; ++nb) {
            const uint32_t q = qcol + nb * 16;
            tau_l[nb] = q < a.n_queries ? a.tau[q] : __builtin_inff();
        }
#pragma unroll
        for (int rg = 0; rg < kRGI; ++rg) {
            const uint32_t r = row0 + rg * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int nb = 0; nb < kNBI; ++nb) {
                const f32x4 v = acc[rg][nb];
                const float t = tau_l[nb];
                if (v[0] >= t || v[1] >= t || v[2] >= t || v[3] >= t) {
                    const uint32_t q = qcol + nb * 16;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (v[i] >= t && r + i <= last_row) {
                            const uint32_t slot = atomicAdd(&a.st[q].n_cand, 1u);
                            if (slot < a.st[q].cap)
                                a.cand[static_cast<size_t>(q) * a.cand_stride + slot] = pack_result(v[i], r + i);
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The batched main kernel: 8-phase LDS-DMA GEMM over the image (the guide's 256 x 256 x 64 structure).
//
// Workgroup = 8 waves as 2 (rows) x 4 (queries); tile = 256 rows x 256 queries; a wave owns 128 x 64 = 32 accumulator
// tiles of v_mfma_f32_16x16x32_f16.  BOTH operands reach LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR hop): a
// K-tile (64 k) is four 16 KiB half-tiles consumed in the order A0, B0, B1, A1 (A half h = rows h*128..+127 of the
// tile, B half = queries hq*128..+127); half-tile s of the workgroup's stream lives in slot s & 7 and is staged 7
// phases ahead by two DMAs per thread.  Phase p of a K-tile multiplies quadrant (A0,B0), (A0,B1), (A1,B1), (A1,B0):
// 16 MFMAs per wave between two raw s_barriers.  `s_waitcnt vmcnt(6)` once per K-tile (phase 3, after that phase's
// DMAs were issued) retires everything the next K-tile reads and leaves three half-tiles in flight -- never
// __syncthreads() in the loop, its fence would drain the DMAs with vmcnt(0).  Both layouts are already fragment-major
// (image: [tile][K-chunk][row block 16][k-step 2][lane]; queries: [qblock][k-step][col block 16][lane]), so every DMA
// destination and every ds_read_b128 is lane-linear: no swizzle, no bank conflicts.
//   RAW: a slot is read one phase after the vmcnt + barrier that retires it.  WAR: a slot is restaged >= 1 phase after
//   its last ds_read, behind the reading phase's lgkmcnt(0) and closing barrier.  Extra vector-memory operations in the
//   stream (threshold DMA, candidate stores, atomics) are OLDER than the three half-tiles a wait leaves in flight, so
//   they only make a wait longer, never shorter.
// Persistent: the grid is one workgroup per CU; a workgroup walks "units" = (row tile, query block of 256).  The units
// of one row tile (n_queries > 256) are consecutive entries of ONE XCD's work list and run on adjacent workgroups of
// that XCD at the same time, so the tile's second..fourth read hits that XCD's L2 (speed only, never correctness).
// Epilogue (filter mode): thresholds of the unit arrive by a 4-byte LDS-DMA at its first phase; an accumulator that
// passes is appended to a workgroup-local LDS list (ds_add_rtn: no global atomic, no vmcnt drain, in the loop); the
// list is flushed to the per-query candidate lists -- one returning global atomic per entry, all lanes at once --
// when it is half full and at the end.  Measured (scratch/gemm_next/gemm8_abl.hip, 256 queries x 10 M x 768): the
// loop is POWER-bound, not issue- or HBM-bound: MFMA alone 2.35 ms at 1.89 GHz, MFMA + LDS reads 3.0 ms at 1.72 GHz,
// with the DMA stream 3.5-3.7 ms at 1.5-1.6 GHz on every schedule variant tried (staggered wave halves, s_setprio,
// role-split rings with 10 phases of HBM lookahead, nt loads): time = energy / power cap.
// ---------------------------------------------------------------------------------------------
constexpr int kHalfBytes = 16384;                 // one half-tile: 128 rows (or queries) x 64 k x 2 B
constexpr uint32_t kG8ListCap = 2048;             // workgroup-local candidate list (entries)
constexpr int kG8OffPack = 8 * kHalfBytes;        // u64[kG8ListCap]
constexpr int kG8OffQid = kG8OffPack + kG8ListCap * 8;   // u32[kG8ListCap]
constexpr int kG8OffTau = kG8OffQid + kG8ListCap * 4;    // f32[2][256]
constexpr int kG8OffCnt = kG8OffTau + 2 * 256 * 4;       // u32
constexpr int kG8LdsBytes = kG8OffCnt + 64;

#define RLR_GLDS16(src, dst)                                                                               \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src),                \
                                     (__attribute__((address_space(3))) void *)(dst), 16, 0, 0)
#define RLR_FENCE() asm volatile("" ::: "memory")

// LDS writes of the epilogue as inline asm: hipcc orders every LDS *store* it can see behind the LDS-DMAs in flight
// with s_waitcnt vmcnt(0) (they might write the same bytes; these never do), which would drain the staging pipeline
// once per unit.  Addresses are byte offsets into LDS.
__device__ __forceinline__ uint32_t lds_add_rtn_u32(uint32_t addr, uint32_t v)
{
    uint32_t r;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr), "v"(v) : "memory");
    return r;
}
__device__ __forceinline__ uint32_t lds_read_b32(uint32_t addr)
{
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ void lds_write_b64(uint32_t addr, uint64_t v)
{
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write_b32(uint32_t addr, uint32_t v)
{
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// VAR (RLR_GEMM8_VARIANT, same-box A/B): bit 0 = s_setprio pair around each MFMA cluster (+2-5 % time: off), bit 1 = no
// sched_barrier behind the phase's lgkmcnt(0) (-1 %), bit 2 = non-temporal DMA for the once-read row half-tiles
// (-2.5 % at 256 queries; only used while a row tile has one reader).  Built: 6 (default) and 0.
template <bool MATERIALISE, int VAR>
__global__ __launch_bounds__(512) void gemm8_kernel(const GemmArgs a, const char *__restrict__ image, uint32_t n_tiles)
{
    __shared__ __attribute__((aligned(1024))) char lds[kG8LdsBytes];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const uint32_t T = a.n_ksteps / 2; // K-tiles of 64 (even: dim % 128 == 0)
    const uint32_t nqb = a.n_qblocks;

    // this workgroup's units: entry s = j + it * J of XCD group (blockIdx % 8)'s list, s -> (tile = (s / nqb) * 8 + xcd, s % nqb)
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3, J = gridDim.x >> 3;
    const uint32_t n_tl = n_tiles > xcd ? (n_tiles - xcd + 7) / 8 : 0;
    const uint32_t n_units = n_tl * nqb;
    const uint32_t n_it = n_units > j ? (n_units - j + J - 1) / J : 0;
    if (n_it == 0)
        return;
    const uint32_t n_phase = n_it * 4 * T; // = half-tiles of this workgroup
    const uint32_t tile_img0 = a.row_begin / kBM;
    const size_t qb_bytes = static_cast<size_t>(a.n_ksteps) * kNB * 1024; // one query block's fragments
    const char *qfrag = reinterpret_cast<const char *>(a.qfrag);

    // staging cursor (wave-uniform): half-tile (s_it, s_t, s_i) of the stream, source bases of unit s_it
    uint32_t s_it = 0, s_t = 0, s_i = 0, s_slot = 0;
    const char *s_abase = nullptr, *s_bbase = nullptr;
    auto set_stage_unit = [&](uint32_t it) {
        const uint32_t s = j + it * J;
        const uint32_t tile = (s / nqb) * 8 + xcd;
        s_abase = image + static_cast<size_t>(tile_img0 + tile) * T * (2 * kHalfBytes);
        s_bbase = qfrag + static_cast<size_t>(s % nqb) * qb_bytes;
    };
    set_stage_unit(0);
    // Sibling re-synchronisation (several query blocks).  The nqb units of one row tile are consecutive list entries, i.e. they run
    // on nqb neighbouring workgroups of this XCD in the same iteration, and the tile's second..n-th read is meant to hit the XCD's L2.
    // Nothing keeps the neighbours in step, though: their drift grows with every unit (the epilogues differ per query block), a line
    // lives ~30 us in the 4 MB L2 at this streaming rate, and rocprofv3 showed the image fetched 1.65 times per batch over 1.6 M rows
    // and 2.85 times over 6.25 M (FETCH_SIZE).  So every kSyncEvery-th unit the group's workgroups meet before staging on: wave 0
    // adds 1 to the group's counter (no-return atomic) and polls it with scalar loads (lgkmcnt, not the vmcnt the DMA ring is counted
    // on) until all nqb have arrived -- or 512 polls have passed (64 were not enough: siblings legitimately differ by a unit's epilogue, and workgroups
    // that gave up let the traffic climb back to 2.1 x): a sibling that never comes (a grid that is not fully resident, e.g. two
    // batched searches on the device at once) costs a few hundred microseconds ONCE -- the workgroup then stops meeting -- never a hang.
    const uint32_t kSyncEvery = a.sync_every ? a.sync_every : 4u;
    const bool sib_sync = a.sync != nullptr && nqb > 1 && nqb <= J && (J % nqb) == 0;
    const uint32_t *sync_word = sib_sync ? a.sync + xcd * 32 + j / nqb : nullptr;
    bool meeting = sib_sync; // (wave-uniform; only wave 0 uses it)
    auto sibling_meet = [&](uint32_t unit) {
        if (lane == 0) // (arrive even after giving up waiting: the siblings that still wait count this workgroup)
            __hip_atomic_fetch_add(const_cast<uint32_t *>(sync_word), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!meeting)
            return;
        const uint32_t want = nqb * (unit / kSyncEvery);
        uint32_t seen = 0;
        for (int tries = 0; tries < 512 && seen < want; ++tries)
            asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(seen) : "s"(sync_word) : "memory");
        meeting = seen >= want;
    };
    auto stage = [&]() {
        const bool is_a = s_i == 0 || s_i == 3;
        char *dst = lds + s_slot * kHalfBytes + wave * 1024;
        // A half: 16 KiB contiguous; B half: k-steps 2t and 2t+1 of col blocks hq*8..+7 = two 8 KiB pieces 16 KiB apart
        const char *src = is_a ? s_abase + (static_cast<size_t>(s_t) * 2 + (s_i == 3)) * kHalfBytes
                               : s_bbase + (static_cast<size_t>(s_t) * 2 * 16 + (s_i == 2) * 8) * 1024;
        const uint32_t second = is_a ? 8192u : 16384u;
        // non-temporal only while a row tile has ONE reader (<= 256 queries).  With several query blocks the units of a tile run
        // side by side on one XCD and their 2nd..n-th read is meant to hit its L2: streamed with `nt` the lines are gone before the
        // siblings arrive -- config 5's share fetched the image 3.2 times per batch (rocprofv3 FETCH_SIZE, profiles/r03_c5_share_*)
        if ((VAR & 4) && is_a && (nqb == 1 || a.a_nt_shared)) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + tid * 16),
                                             (__attribute__((address_space(3))) void *)(dst), 16, 0, 2);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + second + tid * 16),
                                             (__attribute__((address_space(3))) void *)(dst + 8192), 16, 0, 2);
        } else {
            RLR_GLDS16(src + tid * 16, dst);
            RLR_GLDS16(src + second + tid * 16, dst + 8192);
        }
        s_slot = (s_slot + 1) & 7;
        if (++s_i == 4) {
            s_i = 0;
            if (++s_t == T) {
                s_t = 0;
                if (++s_it < n_it) { // past the end: the last unit again, into slots nobody reads any more
                    set_stage_unit(s_it);
                    if (sib_sync && wave == 0 && (s_it % kSyncEvery) == 0)
                        sibling_meet(s_it);
                }
            }
        }
    };

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    acc[h][q][rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    half8 fa[2][4], fb[2][2][2];

    uint32_t *l_cnt = reinterpret_cast<uint32_t *>(lds + kG8OffCnt);
    uint64_t *l_pack = reinterpret_cast<uint64_t *>(lds + kG8OffPack);
    uint32_t *l_qid = reinterpret_cast<uint32_t *>(lds + kG8OffQid);
    const float *l_tau = reinterpret_cast<const float *>(lds + kG8OffTau);
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<size_t>((__attribute__((address_space(3))) char *)lds));
    if (tid == 0)
        *l_cnt = 0;

    // thresholds of unit `it` -> l_tau[it & 1] (4 bytes per lane; waves 4-7 repeat waves 0-3, so every wave issues the
    // same number of vector-memory operations)
    auto tau_dma = [&](uint32_t it, uint32_t ln) {
        if constexpr (!MATERIALISE) {
            const uint32_t s = j + it * J;
            const uint32_t q = min((s % nqb) * kQB + (wave & 3) * 64 + ln, a.n_queries - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.tau + q),
                                             (__attribute__((address_space(3))) void *)(lds + kG8OffTau + (it & 1) * 1024 +
                                                                                       (wave & 3) * 256),
                                             4, 0, 0);
        }
    };

    // prologue: half-tiles 0..6 in flight, then K-tile 0 (0..3) retired
    tau_dma(0, lane);
#pragma unroll
    for (int i = 0; i < 7; ++i)
        stage();
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    RLR_FENCE();

    const half8 *L = reinterpret_cast<const half8 *>(lds);
    constexpr int kSlotH8 = kHalfBytes / 16; // half8 entries per slot
    uint32_t kt = 0, it = 0;                 // K-tile / unit being consumed

#define RLR_READ_A(SLOT)                                                                                     \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int rb = 0; rb < 4; ++rb)         \
        fa[ks][rb] = L[(SLOT) * kSlotH8 + ((wm * 4 + rb) * 2 + ks) * 64 + lane];
#define RLR_READ_B(SLOT, HQ)                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)         \
        fb[HQ][ks][cb] = L[(SLOT) * kSlotH8 + (ks * 8 + wn * 2 + cb) * 64 + lane];
#define RLR_COMPUTE(H, HQ)                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int rb = 0; rb < 4; ++rb)         \
        _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)                                                      \
            acc[H][HQ][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ks][rb], fb[HQ][ks][cb], acc[H][HQ][rb][cb], 0, 0, 0);
#define RLR_PHASE_HEAD(WAIT)                                                                                 \
    stage();                                                                                                 \
    if (WAIT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                              \
    RLR_FENCE();                                                                                             \
    __builtin_amdgcn_s_barrier();                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    if (!(VAR & 2)) __builtin_amdgcn_sched_barrier(0);                                                       \
    if (VAR & 1) __builtin_amdgcn_s_setprio(1);
#define RLR_PHASE_TAIL()                                                                                     \
    if (VAR & 1) __builtin_amdgcn_s_setprio(0);                                                              \
    RLR_FENCE();                                                                                             \
    __builtin_amdgcn_s_barrier();                                                                            \
    RLR_FENCE();
#define RLR_KTILE(KP)                                                                                        \
    RLR_READ_B((KP) * 4 + 1, 0)                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    RLR_READ_A((KP) * 4 + 0)                                                                                 \
    RLR_PHASE_HEAD(false) RLR_COMPUTE(0, 0) RLR_PHASE_TAIL()                                                 \
    RLR_READ_B((KP) * 4 + 2, 1)                                                                              \
    RLR_PHASE_HEAD(false) RLR_COMPUTE(0, 1) RLR_PHASE_TAIL()                                                 \
    RLR_READ_A((KP) * 4 + 3)                                                                                 \
    RLR_PHASE_HEAD(false) RLR_COMPUTE(1, 1) RLR_PHASE_TAIL()                                                 \
    RLR_PHASE_HEAD(true) RLR_COMPUTE(1, 0) RLR_PHASE_TAIL()

    // candidate -> the query's global list (slow path: list overflow, and the flush)
    auto append_global = [&](uint32_t q, uint64_t pk) {
        const uint32_t slot = atomicAdd(&a.st[q].n_cand, 1u);
        if (slot < a.st[q].cap)
            a.cand[static_cast<size_t>(q) * a.cand_stride + slot] = pk;
    };

#pragma unroll 1
    for (uint32_t g = 0; g < n_phase; g += 8) {
        RLR_KTILE(0)
        RLR_KTILE(1)
        kt += 2;
        if (kt == T) { // a unit is complete: consume the accumulators, start the next one
            kt = 0;
            const uint32_t s = j + it * J;
            const uint32_t tile = (s / nqb) * 8 + xcd, qb = s % nqb;
            // everything the epilogue derives from the lane id is recomputed here, once per unit, from an opaque copy:
            // hoisted out of the loop (hipcc does that) those values stay live across the main loop, whose 192
            // accumulator and fragment registers leave no room for them (256 VGPRs + a spill otherwise)
            uint32_t lane_e = static_cast<uint32_t>(lane);
            asm volatile("" : "+v"(lane_e));
            const uint32_t row0 = a.row_begin + tile * kBM + wm * 64 + 4 * (lane_e >> 4);
            const uint32_t last_row = a.row_end - 1;
            const uint32_t q0 = qb * kQB + wn * 32 + (lane_e & 15);
            if constexpr (MATERIALISE) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                            for (int cb = 0; cb < 2; ++cb) {
                                const f32x4 v = acc[h][q][rb][cb];
                                const uint32_t r = row0 + h * 128 + rb * 16;
                                const uint32_t qi = q0 + q * 128 + cb * 16;
                                if (qi < a.n_queries && r <= last_row) {
                                    float *dst = a.scores + static_cast<size_t>(qi) * a.score_stride + (r - a.row_begin);
                                    if (r + 3 <= last_row) {
                                        *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                                    } else {
#pragma unroll
                                        for (int i = 0; i < 4; ++i)
                                            if (r + i <= last_row)
                                                dst[i] = v[i];
                                    }
                                }
                                acc[h][q][rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                            }
            } else {
                float tq[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const uint32_t ql = q * 128 + wn * 32 + cb * 16 + (lane_e & 15);
                        const float t = l_tau[(it & 1) * 256 + ql];
                        tq[q][cb] = qb * kQB + ql < a.n_queries ? t : __builtin_inff();
                    }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                            for (int cb = 0; cb < 2; ++cb) {
                                const f32x4 v = acc[h][q][rb][cb];
                                const float t = tq[q][cb];
                                if (v[0] >= t || v[1] >= t || v[2] >= t || v[3] >= t) {
                                    const uint32_t r = row0 + h * 128 + rb * 16;
                                    const uint32_t qi = q0 + q * 128 + cb * 16;
#pragma unroll
                                    for (int i = 0; i < 4; ++i) {
                                        if (v[i] >= t && r + i <= last_row) {
                                            const uint64_t pk = pack_result(v[i], r + i);
                                            const uint32_t slot = lds_add_rtn_u32(lds_base + kG8OffCnt, 1u);
                                            if (slot < kG8ListCap) {
                                                lds_write_b64(lds_base + kG8OffPack + slot * 8, pk);
                                                lds_write_b32(lds_base + kG8OffQid + slot * 4, qi);
                                            } else {
                                                // list full (> 1024 hits of one unit: a threshold gone wrong): the query is
                                                // marked overflowed and goes back to the single-query pipeline (no-return
                                                // atomic: nothing to wait for)
                                                atomicOr(&a.st[qi].n_cand, 0x80000000u);
                                            }
                                        }
                                    }
                                }
                                acc[h][q][rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                            }
                // flush the workgroup's list when it is half full, and after the last unit
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                RLR_FENCE();
                __builtin_amdgcn_s_barrier();
                RLR_FENCE();
                const uint32_t cnt = lds_read_b32(lds_base + kG8OffCnt);
                if (cnt >= kG8ListCap / 2 || (it + 1 == n_it && cnt > 0)) {
                    const uint32_t n_list = min(cnt, kG8ListCap);
                    for (uint32_t i = tid; i < n_list; i += 512)
                        append_global(l_qid[i], l_pack[i]);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    RLR_FENCE();
                    __builtin_amdgcn_s_barrier();
                    RLR_FENCE();
                    if (tid == 0)
                        lds_write_b32(lds_base + kG8OffCnt, 0u);
                }
            }
            ++it;
            if (it < n_it)
                tau_dma(it, lane_e);
        }
    }
#undef RLR_KTILE
#undef RLR_PHASE_TAIL
#undef RLR_PHASE_HEAD
#undef RLR_COMPUTE
#undef RLR_READ_B
#undef RLR_READ_A
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the tail's dummy DMAs must land before the LDS is handed back
}

// ---------------------------------------------------------------------------------------------
// Single-query scan over the image: the HBM-bound scan of scan.hip at half the bytes.  The image is a
// binary16 copy of the rows, so a query that only needs NOMINATED scores (the exact re-score reads the
// f32 rows of the few candidates) can stream 2 B per element instead of 4.  A wave owns one (tile, wave)
// slot of the image = 32 rows: per K-chunk its four fragments are 4 KB contiguous; a lane holds 8 halfs
// of row (lane & 15) at k-group (lane >> 4), multiplies them with the matching 8 query halfs from LDS
// (v_dot2_f32_f16: exact products, f32 accumulate) and the four k-groups of a row meet in two
// cross-lane adds at the end.  Same outputs as scan.hip: scores[row] and the digit-1 histogram.
// |nominated - reference| <= nomination_eps (both operands rounded to binary16), as in the batched path.
// ---------------------------------------------------------------------------------------------
template <int kScanImgChunks> // K-chunks in flight per wave (4 loads of 1 KiB each)
__global__ __launch_bounds__(256) void scan_image_kernel(const half8 *__restrict__ image, const float *__restrict__ query,
                                                         float *__restrict__ scores, uint32_t *__restrict__ g_hist,
                                                         uint32_t n_rows, uint32_t n_chunks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    half8 *s_q = reinterpret_cast<half8 *>(s_raw);                                   // n_chunks * 8 half8
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_raw + static_cast<size_t>(n_chunks) * 8 * 16);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // query -> binary16 (round to nearest even), 8 consecutive k per entry
    for (uint32_t i = tid; i < n_chunks * 8; i += 256) {
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = static_cast<_Float16>(query[i * 8 + j]);
        s_q[i] = v;
    }
    for (int i = tid; i < kHistBins; i += 256)
        s_hist[i] = 0;
    __syncthreads();

    const uint32_t n_slots = ((n_rows + kBM - 1) / kBM) * 8; // (tile, image wave) pairs, 32 rows each
    const uint32_t n_waves = gridDim.x * 4;
    const int g = lane >> 4;
    for (uint32_t slot = blockIdx.x * 4 + wave; slot < n_slots; slot += n_waves) {
        const uint32_t tile = slot >> 3, w = slot & 7;
        const half8 *ap = image + (static_cast<size_t>(tile) * n_chunks * 8 + w) * 4 * 64 + lane;
        float acc0 = 0.0f, acc1 = 0.0f;
        for (uint32_t c0 = 0; c0 < n_chunks; c0 += kScanImgChunks) {
            half8 x[kScanImgChunks][4];
#pragma unroll
            for (int cc = 0; cc < kScanImgChunks; ++cc) {
                const uint32_t c = min(c0 + cc, n_chunks - 1); // clamped re-read past the end, not used
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    x[cc][f] = __builtin_nontemporal_load(ap + static_cast<size_t>(c) * (8 * 4 * 64) + f * 64);
            }
#pragma unroll
            for (int cc = 0; cc < kScanImgChunks; ++cc) {
                if (c0 + cc < n_chunks) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const half8 q = s_q[((c0 + cc) * 2 + ks) * 4 + g];
                        const half8 a = x[cc][ks];     // row group 0
                        const half8 b = x[cc][2 + ks]; // row group 1
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const half2v qq = {q[2 * j], q[2 * j + 1]};
                            acc0 = __builtin_amdgcn_fdot2(half2v{a[2 * j], a[2 * j + 1]}, qq, acc0, false);
                            acc1 = __builtin_amdgcn_fdot2(half2v{b[2 * j], b[2 * j + 1]}, qq, acc1, false);
                        }
                    }
                }
            }
        }
        // the four k-groups of a row sit 16 lanes apart
        acc0 += __shfl_xor(acc0, 16);
        acc0 += __shfl_xor(acc0, 32);
        acc1 += __shfl_xor(acc1, 16);
        acc1 += __shfl_xor(acc1, 32);
        const uint32_t row = tile * kBM + w * 32 + lane; // lanes 0..31: row groups 0 and 1 back to back
        if (lane < 32 && row < n_rows) {
            const float v = lane < 16 ? acc0 : acc1;
            scores[row] = v;
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

// ---------------------------------------------------------------------------------------------
// Resident-query GEMM over the image: a workgroup keeps ALL K of 64 queries in LDS (dim/32 x 4 KB,
// loaded once) and streams 512 rows past them, so the main loop has no LDS stores and no barriers:
// every wave (64 rows x 64 queries, 16 accumulator tiles) runs its own software pipeline -- A
// fragments straight from the image through a four-chunk register ring (three chunks of lookahead),
// B fragments from LDS one K-step ahead.  The price: the rows are read once per 64-query group
// (groups of one row block are adjacent workgroups of one XCD, so the re-reads hit its L2 -- FETCH_SIZE
// stays at one pass), which is why this kernel only serves batches of <= 128 queries: there it does a
// quarter / half of the 256-wide kernel's MFMA work and the pass is HBM-bound on the binary16 image;
// at 256 queries it measured 7.0 ms against 5.2 ms.
// ---------------------------------------------------------------------------------------------
constexpr int kResQ = 64;               // queries per workgroup
constexpr int kResNB = kResQ / 16;      // column blocks per wave (all waves share the queries)
constexpr int kResRows = 512;           // rows per workgroup: 8 waves x 64
constexpr int kResMaxKsteps = 36;       // dim <= 1152: 36 x 4 KB = 144 KB of the CU's 160 KB LDS
constexpr int kResSlots = 4;            // A ring: chunks in flight per wave

template <bool MATERIALISE>
__global__ __launch_bounds__(512) void gemm_resident_kernel(const GemmArgs a, const half8 *__restrict__ image,
                                                            uint32_t n_qgroups)
{
    __shared__ half8 s_b[kResMaxKsteps * kResNB * 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t bid = blockIdx.x;
    const uint32_t rg512 = (bid / (8 * n_qgroups)) * 8 + (bid & 7); // row group relative to row_begin
    const uint32_t qg = (bid >> 3) % n_qgroups;
    const uint32_t n_rows = a.row_end - a.row_begin;
    if (rg512 * kResRows >= n_rows)
        return;

    // the query group's fragments for every K-step: qfrag is [qblock 256][kstep][colblock 16][lane]
    const half8 *bsrc = a.qfrag + static_cast<size_t>(qg / 4) * a.n_ksteps * kNB * 64 + (qg % 4) * kResNB * 64;
    for (uint32_t i = tid; i < a.n_ksteps * kResNB * 64; i += 512) {
        const uint32_t ks = i / (kResNB * 64), r = i % (kResNB * 64);
        s_b[i] = bsrc[static_cast<size_t>(ks) * kNB * 64 + r];
    }
    __syncthreads();

    const uint32_t row0 = a.row_begin + rg512 * kResRows + wave * 64;
    if (row0 >= a.row_end)
        return; // (no barrier below)
    const uint32_t last_row = a.row_end - 1;
    const uint32_t n_chunks = a.n_ksteps / kKsChunk;
    const uint32_t tile = a.row_begin / kBM + rg512 * 2 + (wave >> 2);
    const half8 *ap = image + (static_cast<size_t>(tile) * n_chunks * 8 + 2 * (wave & 3)) * 4 * 64 + lane;

    f32x4 acc[kRGI][kResNB];
#pragma unroll
    for (int rg = 0; rg < kRGI; ++rg)
#pragma unroll
        for (int nb = 0; nb < kResNB; ++nb)
            acc[rg][nb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    half8 ring[kResSlots][kRGI][kKsChunk];
    const uint32_t last_c = n_chunks - 1;
#define RLR_RES_LOAD_A(CHUNK, SLOT)                                                                \
    _Pragma("unroll") for (int rg = 0; rg < kRGI; ++rg) _Pragma("unroll") for (int ks = 0; ks < kKsChunk; ++ks) \
        ring[SLOT][rg][ks] = ap[static_cast<size_t>(CHUNK) * (8 * 4 * 64) + (rg * 2 + ks) * 64]
#pragma unroll
    for (int sl = 0; sl < kResSlots; ++sl) {
        RLR_RES_LOAD_A(min(static_cast<uint32_t>(sl), last_c), sl);
        // issue order = consumption order: loads return in order, so a younger slot 0 would make the
        // loop's first wait a full drain (hipcc reverses the four slots otherwise)
        __builtin_amdgcn_sched_barrier(0);
    }

    const half8 *sb = s_b + lane;
    half8 bq[2][kResNB];
#pragma unroll
    for (int nb = 0; nb < kResNB; ++nb)
        bq[0][nb] = sb[nb * 64];

#pragma unroll 1
    for (uint32_t c = 0; c < n_chunks; c += kResSlots) {
#pragma unroll
        for (int sl = 0; sl < kResSlots; ++sl) {
            const uint32_t cc = c + sl; // n_chunks is a multiple of kResSlots: no tail, no branch
            {
#pragma unroll
                for (int ks = 0; ks < kKsChunk; ++ks) {
                    // B fragments of the next K-step (clamped re-read at the very end)
                    const uint32_t next = min(cc * kKsChunk + ks + 1, a.n_ksteps - 1);
#pragma unroll
                    for (int nb = 0; nb < kResNB; ++nb)
                        bq[(ks + 1) & 1][nb] = sb[(static_cast<size_t>(next) * kResNB + nb) * 64];
#pragma unroll
                    for (int nb = 0; nb < kResNB; ++nb)
#pragma unroll
                        for (int rg = 0; rg < kRGI; ++rg)
                            acc[rg][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[sl][rg][ks], bq[ks & 1][nb],
                                                                                 acc[rg][nb], 0, 0, 0);
                }
                RLR_RES_LOAD_A(min(cc + kResSlots, last_c), sl);
                __builtin_amdgcn_sched_barrier(0); // keep the refill here: sunk towards its use it is no ring
            }
        }
    }
#undef RLR_RES_LOAD_A
    gemm_image_epilogue<MATERIALISE, kResNB>(a, acc, row0, last_row, qg * kResQ, lane);
}

// ---------------------------------------------------------------------------------------------
// Per-query finish: order the nominated candidates, cut the guard band at the k-th nominated
// score, re-score the band in reference order, order again, emit k.  One workgroup per query.
// status[q]: 0 ok, 1 = candidate or band overflow, 2 = fewer than k candidates (the host then
// runs that query through the single-query pipeline).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kFinCap = 8192;   // candidates per query the finish kernel can order
constexpr uint32_t kBandCap = 2048;  // rows re-scored per query at most

__device__ inline void bitonic_desc_lds(uint64_t *s, uint32_t n_pad, uint32_t nthreads)
{
    for (uint32_t k = 2; k <= n_pad; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n_pad; i += nthreads) {
                const uint32_t ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t x = s[i], y = s[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) {
                        s[i] = y;
                        s[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
}

template <bool F16ROWS>
__global__ __launch_bounds__(256) void batch_finish_kernel(const float4 *__restrict__ rows, uint32_t pitch16, uint32_t dim,
                                                           const float *__restrict__ queries, uint32_t q_pitch,
                                                           const uint64_t *__restrict__ cand, uint32_t cand_stride,
                                                           const SelectState *__restrict__ st, uint32_t k, float two_eps,
                                                           uint64_t *__restrict__ out, uint32_t *__restrict__ status)
{
    __shared__ uint64_t s_c[kFinCap];
    __shared__ uint32_t s_band;
    extern __shared__ __attribute__((aligned(16))) float s_q[];
    const uint32_t q = blockIdx.x;
    const uint32_t n_raw = st[q].n_cand;
    uint64_t *o = out + static_cast<size_t>(q) * k;
    if (n_raw > st[q].cap || n_raw > kFinCap || n_raw < k) {
        if (threadIdx.x == 0)
            status[q] = n_raw < k ? 2u : 1u;
        return;
    }
    for (uint32_t i = threadIdx.x; i < dim; i += 256)
        s_q[i] = queries[static_cast<size_t>(q) * q_pitch + i];
    uint32_t n_pad = 1;
    while (n_pad < n_raw)
        n_pad <<= 1;
    const uint64_t *c = cand + static_cast<size_t>(q) * cand_stride;
    for (uint32_t i = threadIdx.x; i < n_pad; i += 256)
        s_c[i] = i < n_raw ? c[i] : 0ull;
    if (threadIdx.x == 0)
        s_band = 0;
    __syncthreads();
    bitonic_desc_lds(s_c, n_pad, 256);
    // guard band below the k-th nominated score
    const float fk = key_score(static_cast<uint32_t>(s_c[k - 1] >> 32));
    const uint32_t key_lo = score_key(fk - two_eps);
    // The candidates are the rows at or above the floor st.key_lo = (a sample rank's score) - 2 eps.  When that rank is the
    // k-th, the k-th score of all rows cannot lie below it; run_batched takes a higher rank of a smaller sample (the floor
    // then admits ~3 k rows instead of the sample having to be 4 x larger), and the rare query whose k-th nominated score
    // ends up under that rank's score -- its band would reach below the floor, to rows that were never collected -- goes
    // back to the caller.
    if (key_lo < st[q].key_lo) {
        if (threadIdx.x == 0)
            status[q] = 2u;
        return;
    }
    for (uint32_t i = threadIdx.x; i < n_raw; i += 256)
        if (static_cast<uint32_t>(s_c[i] >> 32) >= key_lo)
            atomicMax(&s_band, i + 1);
    __syncthreads();
    const uint32_t band = s_band;
    if (band > kBandCap) {
        if (threadIdx.x == 0)
            status[q] = 1u;
        return;
    }
    // reference-order re-score of the band, one lane per candidate
    for (uint32_t i = threadIdx.x; i < band; i += 256) {
        const uint32_t r = 0xFFFFFFFFu - static_cast<uint32_t>(s_c[i] & 0xFFFFFFFFu);
        const float e = dot_ref_row<F16ROWS>(rows + static_cast<size_t>(r) * pitch16, s_q, dim);
        s_c[i] = pack_result(e, r);
    }
    uint32_t b_pad = 1;
    while (b_pad < band)
        b_pad <<= 1;
    for (uint32_t i = band + threadIdx.x; i < b_pad; i += 256)
        s_c[i] = 0ull;
    __syncthreads();
    bitonic_desc_lds(s_c, b_pad, 256);
    for (uint32_t i = threadIdx.x; i < k; i += 256)
        o[i] = s_c[i];
    if (threadIdx.x == 0)
        status[q] = 0u;
}

// The finish split in three so the re-score runs wide (exact.hip: batch_rescore_kernel) instead of one lane
// per candidate inside the query's single workgroup (446 us -> see DESIGN.md):
//   batch_band_kernel   order the nominated candidates, cut the guard band, leave it in cand[q][0..band)
//   batch_rescore_kernel (exact.hip)  reference-order scores, in place
//   batch_emit_kernel   order the band by the exact scores, emit k
__global__ __launch_bounds__(1024) void batch_band_kernel(uint64_t *__restrict__ cand, uint32_t cand_stride,
                                                          SelectState *__restrict__ st, uint32_t k, float two_eps,
                                                          uint32_t *__restrict__ status)
{
    __shared__ uint64_t s_c[kFinCap];
    __shared__ uint32_t s_hist[2048];
    __shared__ uint32_t s_sel[3];
    __shared__ uint32_t s_band;
    const uint32_t q = blockIdx.x;
    const uint32_t n_raw = st[q].n_cand;
    if (n_raw > st[q].cap || n_raw > kFinCap || n_raw < k) {
        if (threadIdx.x == 0) {
            status[q] = n_raw < k ? 2u : 1u;
            st[q].pad = 0;
        }
        return;
    }
    uint64_t *c = cand + static_cast<size_t>(q) * cand_stride;
    for (uint32_t i = threadIdx.x; i < n_raw; i += 1024)
        s_c[i] = c[i];
    if (threadIdx.x == 0)
        s_band = 0;
    __syncthreads();
    // the guard band below the k-th nominated score: everything with key >= key_lo, in any order (the re-score and
    // the final sort by exact score do not depend on it)
    const float fk = key_score(lds_kth_key(s_c, n_raw, k, s_hist, s_sel, 1024));
    const uint32_t key_lo = score_key(fk - two_eps);
    if (key_lo < st[q].key_lo) { // the band reaches below the floor the candidates were collected at (see batch_finish_kernel)
        if (threadIdx.x == 0) {
            status[q] = 2u;
            st[q].pad = 0;
        }
        return;
    }
    for (uint32_t i = threadIdx.x; i < n_raw; i += 1024) {
        const uint64_t v = s_c[i];
        if (static_cast<uint32_t>(v >> 32) >= key_lo) {
            const uint32_t slot = atomicAdd(&s_band, 1u);
            if (slot < kBandCap)
                c[slot] = v;
        }
    }
    __syncthreads();
    const uint32_t band = s_band;
    if (threadIdx.x == 0) {
        if (band > kBandCap) {
            status[q] = 1u;
            st[q].pad = 0;
        } else {
            st[q].pad = band;
            status[q] = 0u;
        }
    }
}

__global__ __launch_bounds__(1024) void batch_tighten_kernel(uint64_t *__restrict__ cand, uint32_t cand_stride,
                                                             SelectState *__restrict__ st, uint32_t k, float two_eps,
                                                             float *__restrict__ tau)
{
    __shared__ uint64_t s_c[kFinCap];
    __shared__ uint32_t s_keep;
    const uint32_t q = blockIdx.x;
    const uint32_t n_raw = st[q].n_cand;
    if (n_raw > st[q].cap || n_raw > kFinCap || n_raw < k)
        return; // the old threshold stays; the finish flags the query if the list ends up unusable
    uint32_t n_pad = 1;
    while (n_pad < n_raw)
        n_pad <<= 1;
    uint64_t *c = cand + static_cast<size_t>(q) * cand_stride;
    for (uint32_t i = threadIdx.x; i < n_pad; i += 1024)
        s_c[i] = i < n_raw ? c[i] : 0ull;
    if (threadIdx.x == 0)
        s_keep = 0;
    __syncthreads();
    bitonic_desc_lds(s_c, n_pad, 1024);
    const float fk = key_score(static_cast<uint32_t>(s_c[k - 1] >> 32));
    const uint32_t key_lo = score_key(fk - two_eps);
    for (uint32_t i = threadIdx.x; i < n_raw; i += 1024)
        if (static_cast<uint32_t>(s_c[i] >> 32) >= key_lo)
            atomicMax(&s_keep, i + 1);
    __syncthreads();
    const uint32_t keep = s_keep;
    for (uint32_t i = threadIdx.x; i < keep; i += 1024)
        c[i] = s_c[i];
    if (threadIdx.x == 0) {
        st[q].n_cand = keep;
        tau[q] = key_lo == 0 ? -__builtin_inff() : key_score(key_lo);
    }
}

__global__ __launch_bounds__(1024) void batch_emit_kernel(const uint64_t *__restrict__ cand, uint32_t cand_stride,
                                                          const SelectState *__restrict__ st, uint32_t k,
                                                          uint64_t *__restrict__ out)
{
    __shared__ uint64_t s_c[kBandCap];
    const uint32_t q = blockIdx.x;
    const uint32_t band = st[q].pad;
    if (band == 0)
        return; // handed back to the single-query pipeline (status != 0)
    uint32_t b_pad = 1;
    while (b_pad < band)
        b_pad <<= 1;
    const uint64_t *c = cand + static_cast<size_t>(q) * cand_stride;
    for (uint32_t i = threadIdx.x; i < b_pad; i += 1024)
        s_c[i] = i < band ? c[i] : 0ull;
    __syncthreads();
    bitonic_desc_lds(s_c, b_pad, 1024);
    uint64_t *o = out + static_cast<size_t>(q) * k;
    for (uint32_t i = threadIdx.x; i < k; i += 1024)
        o[i] = s_c[i];
}

} // namespace

// upper bound of |nominated score - reference-order dot| for unit-norm operands
float nomination_eps(uint32_t dim, int dtype)
{
    // operand roundings to binary16 (round to nearest even): 2^-11 relative each.  f32 rows:
    // both operands rounded -> 2^-10 (+ cross term); f16 rows are already exact -> 2^-11.
    // Products of two binary16 values are exact in f32; the f32 accumulation (MFMA k-order
    // chain) and the reference's own left-to-right error add (dim + 64) * 2^-24 each; binary16
    // subnormal flushes add < 2 * 2^-25 * sqrt(dim).
    const float op = dtype == RLR_F16 ? 4.8828125e-4f : 9.765625e-4f;
    const float acc = 2.0f * (static_cast<float>(dim) + 64.0f) * 5.9604645e-8f;
    const float sub = 2.0f * 2.9802322e-8f * __builtin_sqrtf(static_cast<float>(dim));
    return (op * 1.001f + acc + sub) * 1.0625f;
}

hipError_t launch_prep_queries(const float *q, uint32_t n_queries, uint32_t q_pitch, uint32_t dim, int dtype,
                               void *qfrag, hipStream_t s)
{
    const uint32_t n_ksteps = dim / 32;
    const uint32_t n_qblocks = (n_queries + kQB - 1) / kQB;
    const size_t total = static_cast<size_t>(n_qblocks) * n_ksteps * kNB * 64 * 8;
    const uint32_t blocks = static_cast<uint32_t>(std::min<size_t>((total + 255) / 256, 4096));
    hipLaunchKernelGGL(prep_queries_kernel, dim3(blocks), dim3(256), 0, s, q, n_queries, q_pitch, dim, n_ksteps,
                       n_qblocks, dtype == RLR_F16 ? 1 : 0, static_cast<_Float16 *>(qfrag));
    return hipGetLastError();
}

// can a batch over `dim`-wide rows run over the nomination image?  (the caller prepares the query fragments in the
// image's natural k order only then)
bool gemm_image_usable(uint32_t dim)
{
    return dim % 128 == 0; // K-tiles of 64, two per loop iteration of gemm8_kernel
}

// one workgroup per CU of the current device, rounded down to a multiple of 8 (the XCD round-robin of the unit lists)
static uint32_t persistent_grid()
{
    static uint32_t cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return 256;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
            n = 256;
        cus[dev] = static_cast<uint32_t>(n) / 8 * 8;
    }
    return cus[dev];
}

hipError_t launch_gemm_nominate(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, uint32_t row_begin,
                                uint32_t row_end, const void *qfrag, uint32_t n_queries, const float *tau,
                                uint64_t *cand, uint32_t cand_stride, SelectState *st, float *scores,
                                size_t score_stride, const void *image, hipStream_t s, uint32_t *sync_ws)
{
    if (row_end <= row_begin)
        return hipSuccess;
    GemmArgs a;
    a.rows = static_cast<const unsigned char *>(rows);
    a.pitch_bytes = pitch16 * 16;
    a.row_begin = row_begin;
    a.row_end = row_end;
    a.n_ksteps = dim / 32;
    a.qfrag = static_cast<const half8 *>(qfrag);
    a.n_qblocks = (n_queries + kQB - 1) / kQB;
    a.n_queries = n_queries;
    a.tau = tau;
    a.cand = cand;
    a.cand_stride = cand_stride;
    a.st = st;
    a.scores = scores;
    a.score_stride = score_stride;
    a.sync = nullptr;
    a.sync_every = 4;
    a.a_nt_shared = 0;
    const uint32_t n_rt = (row_end - row_begin + kBM - 1) / kBM;
    const uint32_t grid = ((n_rt + 7) / 8) * 8 * a.n_qblocks;
    const bool mat = scores != nullptr;
    static const uint32_t resident_max = [] {
        const char *v = getenv("RLR_GEMM_RESIDENT_MAX"); // largest batch served by the resident-query kernel
        return v ? static_cast<uint32_t>(strtoul(v, nullptr, 0)) : 128u;
    }();
    if (image && n_queries <= resident_max && row_begin % kBM == 0 && a.n_ksteps <= kResMaxKsteps &&
        a.n_ksteps % (kKsChunk * kResSlots) == 0) {
        const half8 *img = static_cast<const half8 *>(image);
        const uint32_t n_qg = (n_queries + kResQ - 1) / kResQ;
        const uint32_t n_rg = (row_end - row_begin + kResRows - 1) / kResRows;
        const uint32_t rgrid = ((n_rg + 7) / 8) * 8 * n_qg;
        if (mat)
            hipLaunchKernelGGL((gemm_resident_kernel<true>), dim3(rgrid), dim3(512), 0, s, a, img, n_qg);
        else
            hipLaunchKernelGGL((gemm_resident_kernel<false>), dim3(rgrid), dim3(512), 0, s, a, img, n_qg);
        return hipGetLastError();
    }
    if (image && row_begin % kBM == 0 && a.n_ksteps % 4 == 0) {
        const uint32_t n_units = n_rt * a.n_qblocks;
        const uint32_t g8 = std::max<uint32_t>(8, std::min<uint32_t>(persistent_grid(), (n_units + 7) / 8 * 8));
        const char *img = static_cast<const char *>(image);
        static const bool no_sync = getenv("RLR_GEMM8_NO_SIBLING_SYNC") != nullptr;
        // sibling groups only exist on a grid of one workgroup per CU (all of them resident at once)
        if (sync_ws && a.n_qblocks > 1 && g8 == persistent_grid() && !no_sync) {
            const hipError_t e = hipMemsetAsync(sync_ws, 0, 256 * sizeof(uint32_t), s);
            if (e != hipSuccess)
                return e;
            a.sync = sync_ws;
            static const uint32_t every = getenv("RLR_GEMM8_SYNC_EVERY") ? static_cast<uint32_t>(std::max(1, atoi(getenv("RLR_GEMM8_SYNC_EVERY")))) : 4u;
            a.sync_every = every;
        }
        static const bool a_nt = getenv("RLR_GEMM8_A_NT") != nullptr;
        a.a_nt_shared = a_nt ? 1u : 0u;
        static const int var = [] {
            const char *v = getenv("RLR_GEMM8_VARIANT");
            return v ? static_cast<int>(strtol(v, nullptr, 0)) : 6; // same-box A/B: 6 is 2-4 % faster than 0, 1 is slower
        }();
#define RLR_G8(VAR)                                                                                           \
    if (mat)                                                                                                  \
        hipLaunchKernelGGL((gemm8_kernel<true, VAR>), dim3(g8), dim3(512), 0, s, a, img, n_rt);               \
    else                                                                                                      \
        hipLaunchKernelGGL((gemm8_kernel<false, VAR>), dim3(g8), dim3(512), 0, s, a, img, n_rt)
        if (var == 0) {
            RLR_G8(0);
        } else {
            RLR_G8(6);
        }
#undef RLR_G8
        return hipGetLastError();
    }
    if (dtype == RLR_F16) {
        if (mat)
            hipLaunchKernelGGL((gemm_nominate_kernel<true, true>), dim3(grid), dim3(512), 0, s, a);
        else
            hipLaunchKernelGGL((gemm_nominate_kernel<true, false>), dim3(grid), dim3(512), 0, s, a);
    } else {
        if (mat)
            hipLaunchKernelGGL((gemm_nominate_kernel<false, true>), dim3(grid), dim3(512), 0, s, a);
        else
            hipLaunchKernelGGL((gemm_nominate_kernel<false, false>), dim3(grid), dim3(512), 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_build_image(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, uint32_t n_rows,
                              uint32_t tile_begin, uint32_t tile_end, void *image, hipStream_t s)
{
    if (tile_end <= tile_begin)
        return hipSuccess;
    const uint32_t n_chunks = dim / 64;
    const size_t total = static_cast<size_t>(tile_end - tile_begin) * n_chunks * 8 * 4 * 64;
    const uint32_t blocks = static_cast<uint32_t>(std::min<size_t>((total + 255) / 256, 65536));
    const unsigned char *r = static_cast<const unsigned char *>(rows);
    if (dtype == RLR_F16)
        hipLaunchKernelGGL(build_image_kernel<true>, dim3(blocks), dim3(256), 0, s, r, pitch16 * 16, n_rows, n_chunks,
                           tile_begin, tile_end, static_cast<half8 *>(image));
    else
        hipLaunchKernelGGL(build_image_kernel<false>, dim3(blocks), dim3(256), 0, s, r, pitch16 * 16, n_rows, n_chunks,
                           tile_begin, tile_end, static_cast<half8 *>(image));
    return hipGetLastError();
}

hipError_t launch_scan_image(const void *image, uint32_t n_rows, uint32_t dim, const float *query, float *scores,
                             uint32_t *hist, int n_cu, hipStream_t s)
{
    if (n_rows == 0)
        return hipSuccess;
    const uint32_t n_chunks = dim / 64;
    const uint32_t n_slots = ((n_rows + kBM - 1) / kBM) * 8;
    // 4 chunks in flight x ONE workgroup per CU: 6.65-6.96 TB/s at 10 M x 768 (three interleaved repeats; 3 chunks x 4 workgroups,
    // the round-2 setting: 6.28-6.34 on the same box; scratch/sweep_img2.sh) -- like the f32 scan, one wave per SIMD streams better
    // than four (scan.hip, plan_scan); round 2 had only swept 2-16 workgroups per CU (6.25-6.43, scratch/sweep_img.sh)
    const uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>((n_slots + 3) / 4, static_cast<uint32_t>(n_cu)));
    const size_t lds = static_cast<size_t>(n_chunks) * 8 * 16 + kHistBins * sizeof(uint32_t);
    static const int tune = [] {
        const char *v = getenv("RLR_SCAN_IMAGE_VARIANT"); // chunks in flight | workgroups per CU << 8
        return v ? static_cast<int>(strtol(v, nullptr, 0)) : 0;
    }();
    const int nc = (tune & 0xFF) ? (tune & 0xFF) : 4;
    uint32_t grid = blocks;
    if ((tune >> 8) & 0xFF)
        grid = std::max<uint32_t>(1, std::min<uint32_t>((n_slots + 3) / 4, static_cast<uint32_t>(n_cu) * ((tune >> 8) & 0xFF)));
    const half8 *img = static_cast<const half8 *>(image);
    switch (nc) {
    case 2: hipLaunchKernelGGL(scan_image_kernel<2>, dim3(grid), dim3(256), lds, s, img, query, scores, hist, n_rows, n_chunks); break;
    case 4: hipLaunchKernelGGL(scan_image_kernel<4>, dim3(grid), dim3(256), lds, s, img, query, scores, hist, n_rows, n_chunks); break;
    case 6: hipLaunchKernelGGL(scan_image_kernel<6>, dim3(grid), dim3(256), lds, s, img, query, scores, hist, n_rows, n_chunks); break;
    default: hipLaunchKernelGGL(scan_image_kernel<3>, dim3(grid), dim3(256), lds, s, img, query, scores, hist, n_rows, n_chunks); break;
    }
    return hipGetLastError();
}

size_t image_bytes(uint32_t dim, uint64_t n_rows)
{
    const uint64_t tiles = (n_rows + kBM - 1) / kBM;
    return static_cast<size_t>(tiles) * kBM * dim * 2;
}

hipError_t launch_batch_finish(const void *rows, uint32_t pitch16, uint32_t dim, int dtype, const float *queries,
                               uint32_t q_pitch, uint32_t n_queries, uint64_t *cand, uint32_t cand_stride,
                               SelectState *st, uint32_t k, float two_eps, uint64_t *out, uint32_t *status,
                               hipStream_t s)
{
    static const bool fused = [] {
        const char *v = getenv("RLR_BATCH_FINISH_FUSED"); // A/B switch: the one-kernel finish
        return v && v[0] == '1';
    }();
    if (!fused && batch_rescore_fits(pitch16, dim, dtype)) {
        hipLaunchKernelGGL(batch_band_kernel, dim3(n_queries), dim3(1024), 0, s, cand, cand_stride, st, k, two_eps, status);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return e;
        if (!launch_batch_rescore(rows, pitch16, dim, dtype, queries, q_pitch, n_queries, cand, cand_stride, st, s, &e))
            return hipErrorInvalidValue; // cannot happen: batch_rescore_fits said yes
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL(batch_emit_kernel, dim3(n_queries), dim3(1024), 0, s, cand, cand_stride, st, k, out);
        return hipGetLastError();
    }
    // rows too large for the staged layout (or RLR_BATCH_FINISH_FUSED=1): the one-kernel finish
    const size_t lds = static_cast<size_t>(dim) * sizeof(float);
    const float4 *r4 = static_cast<const float4 *>(rows);
    if (dtype == RLR_F16)
        hipLaunchKernelGGL(batch_finish_kernel<true>, dim3(n_queries), dim3(256), lds, s, r4, pitch16, dim, queries,
                           q_pitch, cand, cand_stride, st, k, two_eps, out, status);
    else
        hipLaunchKernelGGL(batch_finish_kernel<false>, dim3(n_queries), dim3(256), lds, s, r4, pitch16, dim, queries,
                           q_pitch, cand, cand_stride, st, k, two_eps, out, status);
    return hipGetLastError();
}

hipError_t launch_batch_tighten(uint64_t *cand, uint32_t cand_stride, SelectState *st, uint32_t n_queries, uint32_t k,
                                float two_eps, float *tau, hipStream_t s)
{
    hipLaunchKernelGGL(batch_tighten_kernel, dim3(n_queries), dim3(1024), 0, s, cand, cand_stride, st, k, two_eps, tau);
    return hipGetLastError();
}

uint32_t batch_finish_capacity()
{
    return kFinCap;
}

} // namespace rlr
