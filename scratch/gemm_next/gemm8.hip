// Prototype of the 8-phase LDS-DMA nomination GEMM (scratch/gemm_next/PLAN.md): standalone, own layouts, not linked
// into the library.  C[q][row] = sum_k A[row][k] * B[q][k], binary16 operands, f32 accumulation, 256 queries.
//
//   hipcc --offload-arch=gfx950 -O3 -o gemm8 gemm8.hip && ./gemm8            (check on 1024 rows, then time 10 M x 768)
//
// Workgroup = 512 threads = 8 waves as 2 (rows) x 4 (queries), tile 256 rows x 256 queries x BK 64, persistent over
// row tiles.  LDS = 8 slots x 16 KiB; a K-tile is four half-tiles consumed in the order A0, B0, B1, A1 (A half h =
// rows h*128..+127 of the tile, B half = queries hq*128..+127); half-tile s of the workgroup's stream lives in slot
// s & 7 and is staged 7 phases ahead by two global_load_lds_dwordx4 per thread.  Phase p of a K-tile multiplies
// quadrant (A0,B0), (A0,B1), (A1,B1), (A1,B0); `s_waitcnt vmcnt(6)` once per K-tile (phase 3, after that phase's
// DMAs were issued) retires everything the next K-tile reads and leaves three half-tiles in flight.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHalfBytes = 16384;

#define GLDS(src, dst)                                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src),                \
                                     (__attribute__((address_space(3))) void *)(dst), 16, 0, 0)
#define FENCE() asm volatile("" ::: "memory")

// image layouts (both operands fragment-major, so the DMA destination and every ds_read_b128 are lane-linear):
//   A: [row tile][K-tile][half 2][k-step 2][row block 8][lane 64][8 halfs]   row  = tile*256 + h*128 + rb*16 + (lane&15)
//   B:           [K-tile][half 2][k-step 2][col block 8][lane 64][8 halfs]   query = hq*128 + cb*16 + (lane&15)
//   k = ktile*64 + ks*32 + (lane>>4)*8 + j
// wave (wm, wn) multiplies row blocks wm*4..+3 of each A half with col blocks wn*2..+1 of each B half.

template <bool MATERIALISE>
__global__ __launch_bounds__(512) void gemm8_kernel(const char *__restrict__ A, const char *__restrict__ B, uint32_t n_tiles,
                                                    uint32_t T, float tau, float *__restrict__ C, uint32_t ldc,
                                                    unsigned *__restrict__ count)
{
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
        const char *src = is_a ? A + ((static_cast<size_t>(tile) * T + s_t) * 2 + (s_i == 3)) * kHalfBytes
                               : B + (static_cast<size_t>(s_t) * 2 + (s_i == 2)) * kHalfBytes;
        char *dst = lds + s_slot * kHalfBytes + wave * 1024;
        GLDS(src + tid * 16, dst);
        GLDS(src + 8192 + tid * 16, dst + 8192);
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
        a[ks][rb] = L[(SLOT) * kSlotH8 + (ks * 8 + wm * 4 + rb) * 64 + lane];
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
                                } else {
                                    passed += v > tau;
                                }
                            }
                            acc[h][q][rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the tail's dummy DMAs must land before the LDS is handed back
    if (!MATERIALISE && passed)
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

static size_t a_index(uint32_t T, uint32_t row, uint32_t k)
{
    const uint32_t tile = row / 256, r = row % 256, h = r / 128, rb = (r % 128) / 16, l15 = r % 16;
    const uint32_t t = k / 64, ks = (k % 64) / 32, kg = (k % 32) / 8, j = k % 8;
    return ((((static_cast<size_t>(tile) * T + t) * 2 + h) * 2 + ks) * 8 + rb) * 512 + (kg * 16 + l15) * 8 + j;
}
static size_t b_index(uint32_t q, uint32_t k)
{
    const uint32_t hq = q / 128, cb = (q % 128) / 16, l15 = q % 16;
    const uint32_t t = k / 64, ks = (k % 64) / 32, kg = (k % 32) / 8, j = k % 8;
    return (((static_cast<size_t>(t) * 2 + hq) * 2 + ks) * 8 + cb) * 512 + (kg * 16 + l15) * 8 + j;
}

int main(int argc, char **argv)
{
    const uint32_t K = 768, T = K / 64, NQ = 256;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kHalfBytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kHalfBytes));
    // ---- check: 5 row tiles on 2 workgroups (uneven split: 3 + 2 tiles), against double arithmetic on the host
    {
        const uint32_t n = 5 * 256, tiles = n / 256;
        std::vector<_Float16> ha(static_cast<size_t>(n) * K), hb(static_cast<size_t>(NQ) * K);
        std::vector<float> fa(ha.size()), fb(hb.size());
        srand(7);
        for (uint32_t r = 0; r < n; ++r)
            for (uint32_t k = 0; k < K; ++k) {
                const _Float16 v = static_cast<_Float16>((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
                ha[a_index(T, r, k)] = v;
                fa[static_cast<size_t>(r) * K + k] = static_cast<float>(v);
            }
        for (uint32_t q = 0; q < NQ; ++q)
            for (uint32_t k = 0; k < K; ++k) {
                const _Float16 v = static_cast<_Float16>((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
                hb[b_index(q, k)] = v;
                fb[static_cast<size_t>(q) * K + k] = static_cast<float>(v);
            }
        char *dA, *dB; float *dC;
        CK(hipMalloc(&dA, ha.size() * 2)); CK(hipMalloc(&dB, hb.size() * 2)); CK(hipMalloc(&dC, static_cast<size_t>(NQ) * n * 4));
        CK(hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
        int bad_total = 0;
        for (int rep = 0; rep < 20; ++rep) { // the race screen the guide asks for: many runs, same answer
            CK(hipMemset(dC, 0xFF, static_cast<size_t>(NQ) * n * 4));
            hipLaunchKernelGGL(gemm8_kernel<true>, dim3(2), dim3(512), 8 * kHalfBytes, 0, dA, dB, tiles, T, 0.0f, dC, n, nullptr);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            std::vector<float> c(static_cast<size_t>(NQ) * n);
            CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
            int bad = 0; double worst = 0;
            for (uint32_t q = 0; q < NQ; q += (rep == 0 ? 1 : 7))
                for (uint32_t r = 0; r < n; r += (rep == 0 ? 1 : 5)) {
                    double ex = 0;
                    for (uint32_t k = 0; k < K; ++k) ex += static_cast<double>(fa[static_cast<size_t>(r) * K + k]) * fb[static_cast<size_t>(q) * K + k];
                    const double d = std::fabs(ex - c[static_cast<size_t>(q) * n + r]);
                    worst = std::max(worst, d);
                    if (!(d <= 1e-4)) { if (bad < 3) printf("  q %u r %u got %g want %g\n", q, r, c[static_cast<size_t>(q) * n + r], ex); ++bad; }
                }
            if (rep == 0 || bad) printf("check rep %d: mismatches %d, worst |diff| %.3g\n", rep, bad, worst);
            bad_total += bad;
        }
        printf("check: %s\n", bad_total ? "FAILED" : "ok (20 runs)");
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
        if (bad_total) return 1;
    }
    // ---- time: 10 M rows x 768, 256 queries, one workgroup per CU
    {
        const uint32_t tiles = argc > 1 ? static_cast<uint32_t>(atoi(argv[1])) : 39062;
        const size_t a_elems = static_cast<size_t>(tiles) * 256 * K, b_elems = static_cast<size_t>(NQ) * K;
        _Float16 *dA, *dB; unsigned *dCount;
        CK(hipMalloc(&dA, a_elems * 2)); CK(hipMalloc(&dB, b_elems * 2)); CK(hipMalloc(&dCount, 4));
        hipLaunchKernelGGL(fill_half_kernel, dim3(4096), dim3(256), 0, 0, dA, a_elems, 1u);
        hipLaunchKernelGGL(fill_half_kernel, dim3(64), dim3(256), 0, 0, dB, b_elems, 2u);
        CK(hipMemset(dCount, 0, 4));
        CK(hipDeviceSynchronize());
        hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
        const uint32_t grid = std::min<uint32_t>(tiles, prop.multiProcessorCount);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(gemm8_kernel<false>, dim3(grid), dim3(512), 8 * kHalfBytes, 0, reinterpret_cast<const char *>(dA),
                               reinterpret_cast<const char *>(dB), tiles, T, 0.5f, nullptr, 0, dCount);
            CK(hipGetLastError());
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double flop = 2.0 * tiles * 256.0 * NQ * K;
            printf("time rep %d: %.3f ms  %.3f PFLOP/s  %.2f TB/s of A\n", rep, ms, flop / ms / 1e12, a_elems * 2.0 / ms / 1e9);
        }
        unsigned cnt = 0; CK(hipMemcpy(&cnt, dCount, 4, hipMemcpyDeviceToHost));
        printf("passed threshold: %u\n", cnt);
    }
    return 0;
}
