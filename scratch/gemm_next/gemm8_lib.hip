// The 8-phase LDS-DMA nomination GEMM of gemm8.hip (verified there) re-addressed for the LIBRARY's operand layouts --
// the nomination image of csrc/gemm.hip and the query fragments of launch_prep_queries -- and checked against the
// library's own launch_gemm_nominate (materialised scores).  NOT RUN YET: written after the round's GPU budget was
// spent; it compiles, the schedule is byte for byte the verified prototype's, only the addresses differ.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o /tmp/gemm8_lib scratch/gemm_next/gemm8_lib.hip \
//         -L rust-local-rag_amd -lrlr_gpu -Wl,-rpath,$PWD/rust-local-rag_amd && /tmp/gemm8_lib
//
// Both layouts already fit: a (tile, K-chunk) block of the image is 32 KiB ordered [row block 16][k-step 2][lane], so
// A half h (rows h*128..+127) is its contiguous 16 KiB half; the query fragments are [k-step][col block 16][lane], so B
// half hq of a K-tile is two contiguous 8 KiB pieces (one per k-step) -- one DMA each.
//
// Workgroup = 512 threads = 8 waves as 2 (rows) x 4 (queries), tile 256 rows x 256 queries x BK 64, persistent over
// row tiles.  LDS = 8 slots x 16 KiB; a K-tile is four half-tiles consumed in the order A0, B0, B1, A1 (A half h =
// rows h*128..+127 of the tile, B half = queries hq*128..+127); half-tile s of the workgroup's stream lives in slot
// s & 7 and is staged 7 phases ahead by two global_load_lds_dwordx4 per thread.  Phase p of a K-tile multiplies
// quadrant (A0,B0), (A0,B1), (A1,B1), (A1,B0); `s_waitcnt vmcnt(6)` once per K-tile (phase 3, after that phase's
// DMAs were issued) retires everything the next K-tile reads and leaves three half-tiles in flight.
#include <hip/hip_runtime.h>
#include "../../rust-local-rag_amd/csrc/common.h"
#include "../../rust-local-rag_amd/csrc/kernels.h"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <functional>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHalfBytes = 16384;

#define GLDS(src, dst)                                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src),                \
                                     (__attribute__((address_space(3))) void *)(dst), 16, 0, 0)
#define FENCE() asm volatile("" ::: "memory")

// library layouts:
//   A image: [row tile][K-chunk][row block 16][k-step 2][lane 64][8 halfs]   row = tile*256 + rbk*16 + (lane&15)
//   B frags: [k-step dim/32][col block 16][lane 64][8 halfs]                 query = cb*16 + (lane&15)
//   k = kstep*32 + (lane>>4)*8 + j (natural order: queries prepared with dtype f16, as run_batched does for the image)
// LDS half-tile images: A [row block 8][k-step 2][lane] (a linear copy), B [k-step 2][col block 8][lane].
// wave (wm, wn) multiplies row blocks wm*4..+3 of each A half with col blocks wn*2..+1 of each B half.

// MODE 0: count scores above tau (timing); 1: materialise C[q][row]; 2: filter -- append (score, row) >= tau_q[q] to
// cand[q] exactly as the library's gemm_epilogue does (per-lane atomic slot, pack_result)
template <int MODE>
__global__ __launch_bounds__(512) void gemm8_kernel(const char *__restrict__ A, const char *__restrict__ B, uint32_t n_tiles,
                                                    uint32_t T, float tau, float *__restrict__ C, uint32_t ldc,
                                                    unsigned *__restrict__ count, const float *__restrict__ tau_q,
                                                    uint64_t *__restrict__ cand, uint32_t cand_stride,
                                                    rlr::SelectState *__restrict__ st, uint32_t n_rows)
{
    constexpr bool MATERIALISE = MODE == 1;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const uint32_t n_it = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x; // host: gridDim.x <= n_tiles
    const uint32_t n_phase = n_it * 4 * T;                                      // = half-tiles of this workgroup

    // staging cursor (all wave-uniform): half-tile s = (it_s, t_s, i_s)
    uint32_t s_it = 0, s_t = 0, s_i = 0, s_slot = 0;
    auto stage = [&]() {
        const uint32_t it = s_it < n_it ? s_it : n_it - 1; // past the end: re-read into a slot nobody reads again
        const uint32_t tile = blockIdx.x + it * gridDim.x;
        const bool is_a = s_i == 0 || s_i == 3;
        char *dst = lds + s_slot * kHalfBytes + wave * 1024;
        // A half: 16 KiB contiguous; B half: k-steps 2t and 2t+1 of col blocks hq*8..+7 = two 8 KiB pieces 16 KiB apart
        const char *src = is_a ? A + ((static_cast<size_t>(tile) * T + s_t) * 2 + (s_i == 3)) * kHalfBytes
                               : B + (static_cast<size_t>(s_t) * 2 * 16 + (s_i == 2) * 8) * 1024;
        const uint32_t second = is_a ? 8192u : 16384u;
        GLDS(src + tid * 16, dst);
        GLDS(src + second + tid * 16, dst + 8192);
        s_slot = (s_slot + 1) & 7;
        if (++s_i == 4) {
            s_i = 0;
            if (++s_t == T) {
                s_t = 0;
                ++s_it;
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
    half8 a[2][4], b[2][2][2];
    unsigned passed = 0;
    // filter mode: this lane's four query columns' thresholds, loaded and waited for BEFORE the first DMA (an
    // ordinary load result used while DMAs are in flight would make hipcc wait vmcnt(0) in the loop)
    float tq[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    if constexpr (MODE == 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                tq[q][cb] = tau_q[q * 128 + wn * 32 + cb * 16 + (lane & 15)];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                asm volatile("" : "+v"(tq[q][cb]));
    }

    // prologue: half-tiles 0..6 in flight, then K-tile 0 (0..3) retired
#pragma unroll
    for (int i = 0; i < 7; ++i)
        stage();
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    FENCE();

    const half8 *L = reinterpret_cast<const half8 *>(lds);
    constexpr int kSlotH8 = kHalfBytes / 16; // half8 entries per slot
    uint32_t kt = 0, it = 0;                 // K-tile / row-tile iteration being consumed

#define READ_A(SLOT)                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int rb = 0; rb < 4; ++rb)  \
        a[ks][rb] = L[(SLOT) * kSlotH8 + ((wm * 4 + rb) * 2 + ks) * 64 + lane];
#define READ_B(SLOT, HQ)                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)  \
        b[HQ][ks][cb] = L[(SLOT) * kSlotH8 + (ks * 8 + wn * 2 + cb) * 64 + lane];
#define COMPUTE(H, HQ)                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int rb = 0; rb < 4; ++rb)  \
        _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)                                 \
            acc[H][HQ][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks][rb], b[HQ][ks][cb], acc[H][HQ][rb][cb], 0, 0, 0);
#define PHASE_HEAD(WAIT)                                  \
    stage();                                              \
    if (WAIT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); \
    FENCE();                                              \
    __builtin_amdgcn_s_barrier();                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_setprio(1);
#define PHASE_TAIL()                                      \
    __builtin_amdgcn_s_setprio(0);                        \
    FENCE();                                              \
    __builtin_amdgcn_s_barrier();                         \
    FENCE();
#define KTILE(KP)                                         \
    READ_B((KP) * 4 + 1, 0)                               \
    __builtin_amdgcn_sched_barrier(0);                    \
    READ_A((KP) * 4 + 0)                                  \
    PHASE_HEAD(false) COMPUTE(0, 0) PHASE_TAIL()          \
    READ_B((KP) * 4 + 2, 1)                               \
    PHASE_HEAD(false) COMPUTE(0, 1) PHASE_TAIL()          \
    READ_A((KP) * 4 + 3)                                  \
    PHASE_HEAD(false) COMPUTE(1, 1) PHASE_TAIL()          \
    PHASE_HEAD(true) COMPUTE(1, 0) PHASE_TAIL()

#pragma unroll 1
    for (uint32_t g = 0; g < n_phase; g += 8) {
        KTILE(0)
        KTILE(1)
        kt += 2;
        if (kt == T) { // a row tile is complete: consume the accumulators, start the next one
            kt = 0;
            const uint32_t tile = blockIdx.x + it * gridDim.x;
            ++it;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float v = acc[h][q][rb][cb][j];
                                if constexpr (MATERIALISE) {
                                    const uint32_t row = tile * 256 + h * 128 + wm * 64 + rb * 16 + 4 * (lane >> 4) + j;
                                    const uint32_t qi = q * 128 + wn * 32 + cb * 16 + (lane & 15);
                                    C[static_cast<size_t>(qi) * ldc + row] = v;
                                } else if constexpr (MODE == 2) {
                                    const uint32_t row = tile * 256 + h * 128 + wm * 64 + rb * 16 + 4 * (lane >> 4) + j;
                                    if (v >= tq[q][cb] && row < n_rows) { // rare: the atomic's return drains the DMAs
                                        const uint32_t qi = q * 128 + wn * 32 + cb * 16 + (lane & 15);
                                        const uint32_t slot = atomicAdd(&st[qi].n_cand, 1u);
                                        if (slot < st[qi].cap)
                                            cand[static_cast<size_t>(qi) * cand_stride + slot] = rlr::pack_result(v, row);
                                    }
                                } else {
                                    passed += v > tau;
                                }
                            }
                            acc[h][q][rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the tail's dummy DMAs must land before the LDS is handed back
    if (MODE == 0 && passed)
        atomicAdd(count, passed);
}

__global__ void fill_half_kernel(_Float16 *p, size_t n, uint32_t seed)
{
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        uint32_t x = static_cast<uint32_t>(i) * 2654435761u ^ seed ^ static_cast<uint32_t>(i >> 32) * 40503u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = static_cast<_Float16>((static_cast<float>(x & 0xFFFF) / 32768.0f - 1.0f) * 0.05f);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv)
{
    using namespace rlr;
    const uint32_t K = 768, T = K / 64, NQ = 256, pitch16 = K * 2 / 16;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kHalfBytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kHalfBytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kHalfBytes));
    std::vector<float> hq(static_cast<size_t>(NQ) * K);
    srand(11);
    for (auto &v : hq) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    float *dQ; void *dQfrag;
    CK(hipMalloc(&dQ, hq.size() * 4));
    CK(hipMemcpy(dQ, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dQfrag, static_cast<size_t>(NQ) * K * 2));
    CK(launch_prep_queries(dQ, NQ, K, K, 1 /* RLR_F16: natural k order, as for the image */, dQfrag, nullptr));
    // ---- check against the library's own kernel: 1200 rows (ragged last tile) on 2 workgroups, 10 runs
    {
        const uint32_t n = 1200, tiles = (n + 255) / 256, ldc = tiles * 256;
        _Float16 *dRows; void *dImg; float *dC, *dRef;
        CK(hipMalloc(&dRows, static_cast<size_t>(n) * K * 2));
        hipLaunchKernelGGL(fill_half_kernel, dim3(256), dim3(256), 0, 0, dRows, static_cast<size_t>(n) * K, 5u);
        CK(hipMalloc(&dImg, image_bytes(K, n)));
        CK(launch_build_image(dRows, pitch16, K, 1, n, 0, tiles, dImg, nullptr));
        CK(hipMalloc(&dC, static_cast<size_t>(NQ) * ldc * 4));
        CK(hipMalloc(&dRef, static_cast<size_t>(NQ) * ldc * 4));
        CK(launch_gemm_nominate(dRows, pitch16, K, 1, 0, n, dQfrag, NQ, nullptr, nullptr, 0, nullptr, dRef, ldc, dImg, nullptr));
        CK(hipDeviceSynchronize());
        std::vector<float> ref(static_cast<size_t>(NQ) * ldc), c(ref.size());
        CK(hipMemcpy(ref.data(), dRef, ref.size() * 4, hipMemcpyDeviceToHost));
        int bad_total = 0;
        for (int rep = 0; rep < 10; ++rep) {
            CK(hipMemset(dC, 0xFF, c.size() * 4));
            hipLaunchKernelGGL(gemm8_kernel<1>, dim3(2), dim3(512), 8 * kHalfBytes, 0, static_cast<const char *>(dImg),
                               static_cast<const char *>(dQfrag), tiles, T, 0.0f, dC, ldc, nullptr, nullptr, nullptr, 0u,
                               nullptr, n);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
            int bad = 0; double worst = 0;
            for (uint32_t q = 0; q < NQ; ++q)
                for (uint32_t r = 0; r < n; ++r) {
                    const double d = std::fabs(static_cast<double>(c[static_cast<size_t>(q) * ldc + r]) - ref[static_cast<size_t>(q) * ldc + r]);
                    worst = std::max(worst, d);
                    if (!(d <= 1e-5)) { if (bad < 3) printf("  q %u r %u got %g want %g\n", q, r, c[static_cast<size_t>(q) * ldc + r], ref[static_cast<size_t>(q) * ldc + r]); ++bad; }
                }
            if (rep == 0 || bad) printf("check rep %d: mismatches %d of %u, worst |diff| %.3g\n", rep, bad, NQ * n, worst);
            bad_total += bad;
        }
        printf("check against launch_gemm_nominate: %s\n", bad_total ? "FAILED" : "ok (10 runs)");
        // filter mode: thresholds midway between each query's 20th and 21st best reference score -> both kernels must
        // collect the same 20 rows per query (order within a list is arbitrary: compare as sets)
        {
            std::vector<float> tau(NQ);
            for (uint32_t q = 0; q < NQ; ++q) {
                std::vector<float> v(ref.begin() + static_cast<size_t>(q) * ldc, ref.begin() + static_cast<size_t>(q) * ldc + n);
                std::sort(v.begin(), v.end(), std::greater<float>());
                tau[q] = 0.5f * (v[19] + v[20]);
            }
            const uint32_t cap = 64;
            float *dTau; uint64_t *dCand[2]; SelectState *dSt[2];
            CK(hipMalloc(&dTau, NQ * 4)); CK(hipMemcpy(dTau, tau.data(), NQ * 4, hipMemcpyHostToDevice));
            std::vector<SelectState> st(NQ);
            memset(st.data(), 0, st.size() * sizeof(SelectState));
            for (auto &x : st) x.cap = cap;
            for (int w = 0; w < 2; ++w) {
                CK(hipMalloc(&dCand[w], static_cast<size_t>(NQ) * cap * 8)); CK(hipMemset(dCand[w], 0, static_cast<size_t>(NQ) * cap * 8));
                CK(hipMalloc(&dSt[w], NQ * sizeof(SelectState)));
                CK(hipMemcpy(dSt[w], st.data(), NQ * sizeof(SelectState), hipMemcpyHostToDevice));
            }
            CK(launch_gemm_nominate(dRows, pitch16, K, 1, 0, n, dQfrag, NQ, dTau, dCand[0], cap, dSt[0], nullptr, 0, dImg, nullptr));
            hipLaunchKernelGGL(gemm8_kernel<2>, dim3(2), dim3(512), 8 * kHalfBytes, 0, static_cast<const char *>(dImg),
                               static_cast<const char *>(dQfrag), tiles, T, 0.0f, nullptr, 0u, nullptr, dTau, dCand[1], cap, dSt[1], n);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            std::vector<uint64_t> hc[2]; std::vector<SelectState> hs[2];
            for (int w = 0; w < 2; ++w) {
                hc[w].resize(static_cast<size_t>(NQ) * cap); hs[w].resize(NQ);
                CK(hipMemcpy(hc[w].data(), dCand[w], hc[w].size() * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hs[w].data(), dSt[w], NQ * sizeof(SelectState), hipMemcpyDeviceToHost));
            }
            int badf = 0;
            for (uint32_t q = 0; q < NQ; ++q) {
                std::vector<uint32_t> rows[2];
                for (int w = 0; w < 2; ++w) {
                    for (uint32_t i = 0; i < std::min(hs[w][q].n_cand, cap); ++i) {
                        float sc; uint32_t r; unpack_result(hc[w][static_cast<size_t>(q) * cap + i], &sc, &r);
                        rows[w].push_back(r);
                    }
                    std::sort(rows[w].begin(), rows[w].end());
                }
                if (rows[0] != rows[1] || rows[0].size() != 20) { if (badf < 3) printf("  filter q %u: library %zu rows, 8-phase %zu rows\n", q, rows[0].size(), rows[1].size()); ++badf; }
            }
            printf("filter-mode candidate sets: %s\n", badf ? "FAILED" : "identical (20 rows per query)");
            bad_total += badf;
        }
        (void)hipFree(dRows); (void)hipFree(dImg); (void)hipFree(dC); (void)hipFree(dRef);
        if (bad_total) return 1;
    }
    // ---- time both kernels in this process: 10 M rows x 768, 256 queries
    {
        const uint32_t n = argc > 1 ? static_cast<uint32_t>(atoi(argv[1])) : 10000000u, tiles = (n + 255) / 256;
        _Float16 *dRows; void *dImg; unsigned *dCount; float *dTau; uint64_t *dCand; SelectState *dSt;
        CK(hipMalloc(&dRows, static_cast<size_t>(n) * K * 2));
        hipLaunchKernelGGL(fill_half_kernel, dim3(4096), dim3(256), 0, 0, dRows, static_cast<size_t>(n) * K, 1u);
        CK(hipMalloc(&dImg, image_bytes(K, n)));
        CK(launch_build_image(dRows, pitch16, K, 1, n, 0, tiles, dImg, nullptr));
        CK(hipMalloc(&dCount, 4)); CK(hipMemset(dCount, 0, 4));
        // library kernel in filter mode with a threshold nothing passes (what its main pass costs)
        std::vector<float> tau(NQ, 1e30f);
        CK(hipMalloc(&dTau, NQ * 4)); CK(hipMemcpy(dTau, tau.data(), NQ * 4, hipMemcpyHostToDevice));
        const uint32_t cap = batch_finish_capacity();
        CK(hipMalloc(&dCand, static_cast<size_t>(NQ) * cap * 8));
        std::vector<SelectState> st(NQ);
        memset(st.data(), 0, st.size() * sizeof(SelectState));
        for (auto &x : st) x.cap = cap;
        CK(hipMalloc(&dSt, NQ * sizeof(SelectState))); CK(hipMemcpy(dSt, st.data(), NQ * sizeof(SelectState), hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
        const uint32_t grid = std::min<uint32_t>(tiles, prop.multiProcessorCount);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const double flop = 2.0 * tiles * 256.0 * NQ * K;
        for (int rep = 0; rep < 8; ++rep) {
            float ms8 = 0, msl = 0;
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(gemm8_kernel<0>, dim3(grid), dim3(512), 8 * kHalfBytes, 0, static_cast<const char *>(dImg),
                               static_cast<const char *>(dQfrag), tiles, T, 1e30f, nullptr, 0, dCount, nullptr, nullptr, 0u,
                               nullptr, n);
            CK(hipGetLastError());
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms8, e0, e1));
            CK(hipEventRecord(e0, 0));
            CK(launch_gemm_nominate(dRows, pitch16, K, 1, 0, n, dQfrag, NQ, dTau, dCand, cap, dSt, nullptr, 0, dImg, nullptr));
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&msl, e0, e1));
            printf("rep %d: 8-phase %.3f ms (%.3f PFLOP/s)   library image kernel %.3f ms (%.3f PFLOP/s)\n", rep, ms8,
                   flop / ms8 / 1e12, msl, flop / msl / 1e12);
        }
    }
    return 0;
}
