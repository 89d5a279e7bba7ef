// Timing ablations of the 8-phase LDS-DMA nomination GEMM (gemm8_lib.hip's kernel, count mode), one process,
// interleaved rounds, in-kernel clock stamps.  The schedule itself is checked in gemm8_lib.hip.
// Variants are flag combinations (F_* below); every variant's count of scores above a threshold is printed so that a
// schedule edit that changes a result shows up (same MFMA order => the counts must be identical).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/gemm8_abl scratch/gemm_next/gemm8_abl.hip && /tmp/gemm8_abl
#include <hip/hip_runtime.h>
#include <algorithm>
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

// variant flags
constexpr int F_PRIO = 1;      // s_setprio pair around the MFMA cluster
constexpr int F_STAG = 2;      // waves 4-7 run one barrier behind waves 0-3 (ping-pong: one half reads while the other multiplies)
constexpr int F_LGKM_EARLY = 4; // s_waitcnt lgkmcnt(0) before the phase's first barrier (required with F_STAG)
constexpr int F_NT_A = 8;      // non-temporal LDS-DMA for the once-read A stream
constexpr int F_NODMA = 16;    // timing only: no DMA in the loop
constexpr int F_SCHED = 32;    // sched_barrier(0) around the MFMA cluster
constexpr int F_NOLDS = 64;    // timing only: no ds_read in the loop (operands read once)
constexpr int F_NOMFMA = 128;  // timing only: no MFMA (the ds_reads are kept alive)
constexpr int F_ASM_ACC = 512; // 4-wave kernel: the 64 accumulator tiles pinned to AGPRs (inline-asm MFMAs with "a" constraints)
constexpr int F_PRIO_HI = 256; // static s_setprio 1 for waves 4-7, no per-cluster flips

template <int ABL>
__global__ __launch_bounds__(512) void gemm8_kernel(const char *__restrict__ A, const char *__restrict__ B, uint32_t n_tiles,
                                                    uint32_t T, float tau, unsigned *__restrict__ count,
                                                    unsigned long long *__restrict__ stamps)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const uint32_t n_it = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const uint32_t n_phase = n_it * 4 * T;

    uint32_t s_it = 0, s_t = 0, s_i = 0, s_slot = 0;
    auto stage = [&]() {
        const uint32_t it = s_it < n_it ? s_it : n_it - 1;
        const uint32_t tile = blockIdx.x + it * gridDim.x;
        const bool is_a = s_i == 0 || s_i == 3;
        char *dst = lds + s_slot * kHalfBytes + wave * 1024;
        const char *src = is_a ? A + ((static_cast<size_t>(tile) * T + s_t) * 2 + (s_i == 3)) * kHalfBytes
                               : B + (static_cast<size_t>(s_t) * 2 * 16 + (s_i == 2) * 8) * 1024;
        const uint32_t second = is_a ? 8192u : 16384u;
        if ((ABL & F_NT_A) && is_a) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + tid * 16),
                                             (__attribute__((address_space(3))) void *)(dst), 16, 0, 2);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + second + tid * 16),
                                             (__attribute__((address_space(3))) void *)(dst + 8192), 16, 0, 2);
        } else {
            GLDS(src + tid * 16, dst);
            GLDS(src + second + tid * 16, dst + 8192);
        }
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

#pragma unroll
    for (int i = 0; i < 7; ++i)
        stage();
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    FENCE();
    if (ABL & F_NODMA) {
        stage();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FENCE();
    }
    if ((ABL & F_STAG) && wm == 1) {
        __builtin_amdgcn_s_barrier();
        FENCE();
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

    const half8 *L = reinterpret_cast<const half8 *>(lds);
    constexpr int kSlotH8 = kHalfBytes / 16;
    uint32_t kt = 0, it = 0;
    if (ABL & F_NOLDS) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
                a[ks][rb] = L[((wm * 4 + rb) * 2 + ks) * 64 + lane];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                b[0][ks][cb] = L[kSlotH8 + (ks * 8 + wn * 2 + cb) * 64 + lane];
                b[1][ks][cb] = L[2 * kSlotH8 + (ks * 8 + wn * 2 + cb) * 64 + lane];
            }
        }
    }
    if ((ABL & F_PRIO_HI) && wave >= 4)
        __builtin_amdgcn_s_setprio(1);

#define READ_A(SLOT)                                                                      \
    if (!(ABL & F_NOLDS)) { _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int rb = 0; rb < 4; ++rb)  \
        a[ks][rb] = L[(SLOT) * kSlotH8 + ((wm * 4 + rb) * 2 + ks) * 64 + lane]; }
#define READ_B(SLOT, HQ)                                                                  \
    if (!(ABL & F_NOLDS)) { _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)  \
        b[HQ][ks][cb] = L[(SLOT) * kSlotH8 + (ks * 8 + wn * 2 + cb) * 64 + lane]; }
#define COMPUTE(H, HQ)                                                                    \
    if (ABL & F_NOMFMA) { _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { _Pragma("unroll") for (int rb = 0; rb < 4; ++rb) asm volatile("" :: "v"(a[ks][rb])); \
        _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) asm volatile("" :: "v"(b[HQ][ks][cb])); } } else                \
    { _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int rb = 0; rb < 4; ++rb)  \
        _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)                                 \
            acc[H][HQ][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks][rb], b[HQ][ks][cb], acc[H][HQ][rb][cb], 0, 0, 0); }
#define PHASE_HEAD(WAIT)                                  \
    if (!(ABL & F_NODMA)) stage();                        \
    if (WAIT && !(ABL & F_NODMA)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); \
    if (ABL & F_LGKM_EARLY) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    FENCE();                                              \
    __builtin_amdgcn_s_barrier();                         \
    if (!(ABL & F_LGKM_EARLY)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    if (ABL & F_SCHED) __builtin_amdgcn_sched_barrier(0); \
    if (ABL & F_PRIO) __builtin_amdgcn_s_setprio(1);
#define PHASE_TAIL()                                      \
    if (ABL & F_PRIO) __builtin_amdgcn_s_setprio(0);      \
    if (ABL & F_SCHED) __builtin_amdgcn_sched_barrier(0); \
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
        if (kt == T) {
            kt = 0;
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
                            for (int j = 0; j < 4; ++j)
                                passed += acc[h][q][rb][cb][j] > tau;
                            acc[h][q][rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((ABL & F_STAG) && wm == 0) {
        FENCE();
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (passed)
        atomicAdd(count, passed);
    if (tid == 0) {
        stamps[blockIdx.x * 2] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Role-split variant: waves 0-3 stage ONLY the A (row) half-tiles, waves 4-7 ONLY the B (query) half-tiles, each into a
// ring of its own (A: 6 slots = 3 K-tiles, B: 4 slots = 2 K-tiles; 160 KiB).  vmcnt is per wave and retires in issue
// order, so in the shared stream every wait for a (fast, L2-resident) B half-tile also waited for the (slow, HBM) A
// half-tile issued just before it: A1 had 4 phases of lookahead.  Here an A half-tile is staged 10-11 phases before its
// first read and a B half-tile 7.  Waves 4-7 run one barrier behind waves 0-3 (F_STAG semantics always on).
//   per K-tile t:  phase 1: G0 stages A0(t+3), waits vmcnt(20) => A1(t) landed;   G1 stages B0(t+2)
//                  phase 2:                                                         G1 stages B1(t+2)
//                  phase 3: G0 stages A1(t+3), waits vmcnt(20) => A0(t+1) landed;  G1 waits vmcnt(8) => B(t+1) landed
//   (4 DMAs per staging wave per half-tile; 20 = 5 younger half-tiles, 8 = 2)
template <int ABL>
__global__ __launch_bounds__(512) void gemm8s_kernel(const char *__restrict__ A, const char *__restrict__ B, uint32_t n_tiles,
                                                     uint32_t T, float tau, unsigned *__restrict__ count,
                                                     unsigned long long *__restrict__ stamps)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool stager_a = wave < 4;
    const int sw = wave & 3;
    const uint32_t n_it = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const uint32_t n_kt = n_it * T;

    uint32_t s_it = 0, s_kt = 0, s_slot = 0; // staging cursor of this wave's role: K-tile (s_it, s_kt), ring slot of its half 0
    auto stage_half = [&](int half) {
        const uint32_t it = s_it < n_it ? s_it : n_it - 1; // past the end: re-read into slots nobody reads again
        const uint32_t tile = blockIdx.x + it * gridDim.x;
        if (stager_a) {
            const char *src = A + ((static_cast<size_t>(tile) * T + s_kt) * 2 + half) * kHalfBytes + sw * 4096 + lane * 16;
            char *dst = lds + (s_slot + half) * kHalfBytes + sw * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (ABL & F_NT_A)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i * 1024),
                                                     (__attribute__((address_space(3))) void *)(dst + i * 1024), 16, 0, 2);
                else
                    GLDS(src + i * 1024, dst + i * 1024);
            }
        } else {
            // B half: k-step 0 = pieces 0..7 at base, k-step 1 = pieces 8..15 at base + 16 KiB
            const char *src = B + (static_cast<size_t>(s_kt) * 2 * 16 + half * 8) * 1024 + (sw >> 1) * 16384 + (sw & 1) * 4096 + lane * 16;
            char *dst = lds + (6 + s_slot + half) * kHalfBytes + sw * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                GLDS(src + i * 1024, dst + i * 1024);
        }
    };
    auto advance = [&]() {
        s_slot += 2;
        if (s_slot == (stager_a ? 6u : 4u))
            s_slot = 0;
        if (++s_kt == T) {
            s_kt = 0;
            ++s_it;
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

    // prologue: A(0..2) / B(0..1) in flight, K-tile 0 retired
    if (stager_a) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            stage_half(0);
            stage_half(1);
            advance();
        }
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            stage_half(0);
            stage_half(1);
            advance();
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    FENCE();
    if (wm == 1) {
        __builtin_amdgcn_s_barrier();
        FENCE();
    }
    if ((ABL & F_PRIO_HI) && wave >= 4)
        __builtin_amdgcn_s_setprio(1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

    const half8 *L = reinterpret_cast<const half8 *>(lds);
    constexpr int kSlotH8 = kHalfBytes / 16;
    uint32_t kt = 0, cA = 0, cB = 0; // consumption cursor: K-tile within the row tile, ring slots of its halves 0

#define S_READ_A(H)                                                                       \
    { const half8 *pa = L + (cA + (H)) * kSlotH8 + lane;                                   \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int rb = 0; rb < 4; ++rb)  \
        a[ks][rb] = pa[((wm * 4 + rb) * 2 + ks) * 64]; }
#define S_READ_B(HQ)                                                                      \
    { const half8 *pb = L + (6 + cB + (HQ)) * kSlotH8 + lane;                              \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)  \
        b[HQ][ks][cb] = pb[(ks * 8 + wn * 2 + cb) * 64]; }
#define S_HEAD()                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    FENCE();                                              \
    __builtin_amdgcn_s_barrier();                         \
    if (ABL & F_SCHED) __builtin_amdgcn_sched_barrier(0); \
    if (ABL & F_PRIO) __builtin_amdgcn_s_setprio(1);

#pragma unroll 1
    for (uint32_t g = 0; g < n_kt; ++g) {
        // phase 0: (A0, B0)
        S_READ_B(0)
        __builtin_amdgcn_sched_barrier(0);
        S_READ_A(0)
        S_HEAD() COMPUTE(0, 0) PHASE_TAIL()
        // phase 1: (A0, B1)
        S_READ_B(1)
        if (stager_a) {
            stage_half(0);
            asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        } else {
            stage_half(0);
        }
        S_HEAD() COMPUTE(0, 1) PHASE_TAIL()
        // phase 2: (A1, B1)
        S_READ_A(1)
        if (!stager_a)
            stage_half(1);
        S_HEAD() COMPUTE(1, 1) PHASE_TAIL()
        // phase 3: (A1, B0)
        if (stager_a) {
            stage_half(1);
            asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        advance();
        S_HEAD() COMPUTE(1, 0) PHASE_TAIL()
        cA = cA == 4 ? 0 : cA + 2;
        cB ^= 2;
        if (++kt == T) {
            kt = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                passed += acc[h][q][rb][cb][j] > tau;
                            acc[h][q][rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (wm == 0) {
        FENCE();
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (passed)
        atomicAdd(count, passed);
    if (tid == 0) {
        stamps[blockIdx.x * 2] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 4-wave variant: one wave per SIMD, wave (wm, wn) owns the 128 x 128 quadrant (A half wm, B half wn) = 64 accumulator
// tiles (256 registers).  LDS fragment reads per K-tile drop from 192 KiB (8 waves x 24) to 128 KiB (4 x 32); one barrier
// per k-step of 64 MFMAs instead of two per 16.  The ring is cut in k-steps (32 k): slot s & 3 = [A0 | A1 | B0 | B1] x
// 8 KiB, every 1-KiB fragment is one DMA piece, wave w stages group w (8 DMAs per thread per k-step), 3 k-steps ahead.
//   per k-step s:  lgkmcnt(0) [fragments of s in registers] -> vmcnt(16) [k-step s+1 landed] -> barrier ->
//                  ds_read fragments of s+1 into the other register set -> stage k-step s+4 into slot s & 3 -> 64 MFMAs
template <int ABL>
__global__ __launch_bounds__(256) void gemm4w_kernel(const char *__restrict__ A, const char *__restrict__ B, uint32_t n_tiles,
                                                     uint32_t T, float tau, unsigned *__restrict__ count,
                                                     unsigned long long *__restrict__ stamps)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const uint32_t n_it = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const uint32_t n_ks = n_it * T * 2; // k-steps of this workgroup

    // staging cursor: k-step (s_it, s_kt, s_ks), ring slot
    uint32_t s_it = 0, s_kt = 0, s_ks = 0, s_slot = 0;
    auto stage = [&]() {
        const uint32_t it = s_it < n_it ? s_it : n_it - 1;
        const uint32_t tile = blockIdx.x + it * gridDim.x;
        char *dst = lds + s_slot * 32768 + wave * 8192;
        const char *src;
        uint32_t stride;
        if (wave < 2) { // A half `wave`: fragments (rb, ks) at (rb * 2 + ks) KiB of the half-tile
            src = A + ((static_cast<size_t>(tile) * T + s_kt) * 2 + wave) * kHalfBytes + s_ks * 1024 + lane * 16;
            stride = 2048;
        } else {        // B half `wave - 2`: fragments of k-step (2 kt + ks), col blocks hq * 8 ..
            src = B + (static_cast<size_t>(s_kt * 2 + s_ks) * 16 + (wave - 2) * 8) * 1024 + lane * 16;
            stride = 1024;
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if ((ABL & F_NT_A) && wave < 2)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * stride),
                                                 (__attribute__((address_space(3))) void *)(dst + p * 1024), 16, 0, 2);
            else
                GLDS(src + p * stride, dst + p * 1024);
        }
        s_slot = (s_slot + 1) & 3;
        if (++s_ks == 2) {
            s_ks = 0;
            if (++s_kt == T) {
                s_kt = 0;
                ++s_it;
            }
        }
    };

    f32x4 acc[8][8];
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
            acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    half8 fa[2][8], fb[2][8];
    unsigned passed = 0;
    const half8 *L = reinterpret_cast<const half8 *>(lds);
    // fragment (group g, block blk) of slot sl: half8 index sl * 2048 + g * 512 + blk * 64 + lane
#define W4_READ(BUF, SL)                                                                  \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                       \
        fa[BUF][i] = L[(SL) * 2048 + wm * 512 + i * 64 + lane];                           \
        fb[BUF][i] = L[(SL) * 2048 + (2 + wn) * 512 + i * 64 + lane];                     \
    }
#define W4_COMPUTE(BUF)                                                                   \
    _Pragma("unroll") for (int rb = 0; rb < 8; ++rb) _Pragma("unroll") for (int cb = 0; cb < 8; ++cb) { \
        if constexpr ((ABL & F_ASM_ACC) != 0)                                             \
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[rb][cb]) : "v"(fa[BUF][rb]), "v"(fb[BUF][cb])); \
        else                                                                              \
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[BUF][rb], fb[BUF][cb], acc[rb][cb], 0, 0, 0); \
    }

    // prologue: k-steps 0..3 in flight; k-step 0 landed, its fragments in register set 0
    stage(); stage(); stage(); stage();
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    FENCE();
    W4_READ(0, 0)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t kt2 = 0; // k-steps consumed of the current row tile

#define W4_STEP(BUF, SL)                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                    \
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                     \
    FENCE();                                                                              \
    __builtin_amdgcn_s_barrier();                                                         \
    FENCE();                                                                              \
    W4_READ((BUF) ^ 1, ((SL) + 1) & 3)                                                    \
    stage();                                                                              \
    if (ABL & F_SCHED) __builtin_amdgcn_sched_barrier(0);                                 \
    W4_COMPUTE(BUF)

#pragma unroll 1
    for (uint32_t s = 0; s < n_ks; s += 4) {
        W4_STEP(0, 0)
        W4_STEP(1, 1)
        W4_STEP(0, 2)
        W4_STEP(1, 3)
        kt2 += 4;
        if (kt2 == 2 * T) {
            kt2 = 0;
#pragma unroll
            for (int rb = 0; rb < 8; ++rb)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        passed += acc[rb][cb][j] > tau;
                    acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (passed)
        atomicAdd(count, passed);
    if (tid == 0) {
        stamps[blockIdx.x * 2] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
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

template <int ABL>
static int run(const char *A, const char *B, uint32_t tiles, uint32_t T, uint32_t grid, unsigned *dCount, unsigned long long *dStamps,
               float *ms, double *ghz)
{
    static bool attr = false;
    if (!attr) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kHalfBytes));
        attr = true;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(gemm8_kernel<ABL>, dim3(grid), dim3(512), 8 * kHalfBytes, 0, A, B, tiles, T, 0.002f, dCount, dStamps);
    CK(hipGetLastError());
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(ms, e0, e1));
    std::vector<unsigned long long> st(grid * 2);
    CK(hipMemcpy(st.data(), dStamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> c(grid);
    for (uint32_t i = 0; i < grid; ++i)
        c[i] = st[2 * i + 1] ? static_cast<double>(st[2 * i]) / static_cast<double>(st[2 * i + 1]) * 0.1 : 0.0; // GHz
    std::sort(c.begin(), c.end());
    *ghz = c[grid / 2];
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 0;
}

template <int ABL>
static int run_s(const char *A, const char *B, uint32_t tiles, uint32_t T, uint32_t grid, unsigned *dCount, unsigned long long *dStamps,
                 float *ms, double *ghz)
{
    static bool attr = false;
    if (!attr) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8s_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 10 * kHalfBytes));
        attr = true;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(gemm8s_kernel<ABL>, dim3(grid), dim3(512), 10 * kHalfBytes, 0, A, B, tiles, T, 0.002f, dCount, dStamps);
    CK(hipGetLastError());
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(ms, e0, e1));
    std::vector<unsigned long long> st(grid * 2);
    CK(hipMemcpy(st.data(), dStamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> c(grid);
    for (uint32_t i = 0; i < grid; ++i)
        c[i] = st[2 * i + 1] ? static_cast<double>(st[2 * i]) / static_cast<double>(st[2 * i + 1]) * 0.1 : 0.0; // GHz
    std::sort(c.begin(), c.end());
    *ghz = c[grid / 2];
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 0;
}

template <int ABL>
static int run_4w(const char *A, const char *B, uint32_t tiles, uint32_t T, uint32_t grid, unsigned *dCount, unsigned long long *dStamps,
                  float *ms, double *ghz)
{
    static bool attr = false;
    if (!attr) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm4w_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
        attr = true;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(gemm4w_kernel<ABL>, dim3(grid), dim3(256), 4 * 32768, 0, A, B, tiles, T, 0.002f, dCount, dStamps);
    CK(hipGetLastError());
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(ms, e0, e1));
    std::vector<unsigned long long> st(grid * 2);
    CK(hipMemcpy(st.data(), dStamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> c(grid);
    for (uint32_t i = 0; i < grid; ++i)
        c[i] = st[2 * i + 1] ? static_cast<double>(st[2 * i]) / static_cast<double>(st[2 * i + 1]) * 0.1 : 0.0; // GHz
    std::sort(c.begin(), c.end());
    *ghz = c[grid / 2];
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 0;
}

int main(int argc, char **argv)
{
    const uint32_t K = 768, T = K / 64, NQ = 256;
    const uint32_t n = argc > 1 ? static_cast<uint32_t>(atoi(argv[1])) : 10000000u, tiles = (n + 255) / 256;
    _Float16 *dA, *dB; unsigned *dCount; unsigned long long *dStamps;
    const size_t a_elems = static_cast<size_t>(tiles) * 256 * K;
    CK(hipMalloc(&dA, a_elems * 2));
    CK(hipMalloc(&dB, static_cast<size_t>(NQ) * K * 2));
    hipLaunchKernelGGL(fill_half_kernel, dim3(4096), dim3(256), 0, 0, dA, a_elems, 1u);
    hipLaunchKernelGGL(fill_half_kernel, dim3(256), dim3(256), 0, 0, dB, static_cast<size_t>(NQ) * K, 7u);
    CK(hipMalloc(&dCount, 4)); CK(hipMemset(dCount, 0, 4));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const uint32_t grid = std::min<uint32_t>(tiles, prop.multiProcessorCount);
    CK(hipMalloc(&dStamps, grid * 16));
    CK(hipDeviceSynchronize());
    const double flop = 2.0 * tiles * 256.0 * NQ * K;
    const char *A = reinterpret_cast<const char *>(dA), *B = reinterpret_cast<const char *>(dB);
    struct V { const char *name; int (*fn)(const char *, const char *, uint32_t, uint32_t, uint32_t, unsigned *, unsigned long long *, float *, double *); };
    const V vs[] = {
        {"plain", run<0>},
        {"plain+ntA", run<F_NT_A>},
        {"stag+prio", run<F_STAG | F_LGKM_EARLY | F_PRIO>},
        {"4w", run_4w<0>},
        {"4w+ntA", run_4w<F_NT_A>},
        {"4w+sched", run_4w<F_SCHED>},
        {"4w+asmacc", run_4w<F_ASM_ACC>},
        {"4w+asmacc+nt", run_4w<F_ASM_ACC | F_NT_A>},
    };
    const int nv = sizeof(vs) / sizeof(vs[0]);
    for (int rep = 0; rep < 5; ++rep) {
        printf("rep %d:\n", rep);
        for (int v = 0; v < nv; ++v) {
            float ms; double ghz; unsigned cnt = 0;
            CK(hipMemset(dCount, 0, 4));
            if (vs[v].fn(A, B, tiles, T, grid, dCount, dStamps, &ms, &ghz)) return 2;
            CK(hipMemcpy(&cnt, dCount, 4, hipMemcpyDeviceToHost));
            printf("   %-12s %.3f ms  %.2f GHz  %.3f PF  %.2f Mcyc  count %u\n", vs[v].name, ms, ghz, flop / ms / 1e12, ms * ghz * 1e3, cnt);
        }
    }
    return 0;
}
