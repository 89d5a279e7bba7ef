// cycles per f32 matrix instruction in a DEPENDENT accumulator chain (what a reference-order dot product is), per variant and per
// number of resident waves per SIMD.   hipcc --offload-arch=gfx950 -O2 -o mfma_f32_rate_probe mfma_f32_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v32f __attribute__((ext_vector_type(32)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ void k(float *out, unsigned long long *cyc, int n)
{
    const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    unsigned long long t0, t1;
    if constexpr (VARIANT == 0) {
        v32f acc;
        for (int i = 0; i < 32; ++i) acc[i] = 0.0f;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, acc, 0, 0, 0);
        t1 = __builtin_readcyclecounter();
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[31];
    } else if constexpr (VARIANT == 1) {
        v16f acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        t1 = __builtin_readcyclecounter();
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[15];
    } else if constexpr (VARIANT == 2) {
        v4f acc = {0, 0, 0, 0};
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n; ++i)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        t1 = __builtin_readcyclecounter();
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[3];
    } else {
        v16f acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n; ++i)
            acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc, 0, 0, 0); // 4 blocks of 16 x 16 x 1
        t1 = __builtin_readcyclecounter();
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[15];
    }
    if (threadIdx.x % 64 == 0)
        cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int V>
void run(const char *name, double macs)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 1 << 20);
    const int n = 4096;
    for (int waves_per_simd : {1, 2, 3}) {
        const int threads = 256 * waves_per_simd; // 4 SIMDs per CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<V><<<256, threads>>>(out, cyc, n); hipDeviceSynchronize();
        hipEventRecord(e0);
        k<V><<<256, threads>>>(out, cyc, n);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256 * 4 * waves_per_simd);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += v; mean /= h.size();
        // s_memtime ticks at a fixed 100 MHz on this part? report both the tick count and the wall-clock rate
        const double tf = macs * 2.0 * n * 256.0 * 4 * waves_per_simd / (ms * 1e-3) / 1e12;
        printf("%-28s %d wave(s)/SIMD: %.3f ms, counter ticks per instruction per wave %.1f, %.1f TFLOP/s whole chip\n", name, waves_per_simd, ms, mean / n, tf);
    }
}

int main()
{
    run<0>("v_mfma_f32_32x32x1_2b_f32", 2048);
    run<1>("v_mfma_f32_32x32x2_f32", 2048);
    run<2>("v_mfma_f32_16x16x4_f32", 1024);
    run<3>("v_mfma_f32_16x16x1_4b_f32", 1024);
    return 0;
}
