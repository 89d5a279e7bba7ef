// What clock does a latency-bound single-wave kernel (the greedy MMR chain, the single-pool Gram) actually run at?
// s_memtime counts shader clocks, s_memrealtime a fixed 100 MHz: their ratio over a dependent-add chain gives the core clock,
// the chain length over the s_memtime delta the cycles per dependent VALU instruction.
//   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <unistd.h>

__global__ void chain_kernel(float *out, unsigned long long *stamps, int n, float x)
{
    float a = static_cast<float>(threadIdx.x);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(x));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamps[0] = t1 - t0;
        stamps[1] = r1 - r0;
    }
}

__global__ void stream_kernel(const float4 *src, float *out, size_t n4)
{
    float s = 0;
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const float4 v = src[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f)
        out[0] = s;
}

static void report(const char *what, unsigned long long *h, int n)
{
    const double clk_mhz = static_cast<double>(h[0]) / static_cast<double>(h[1]) * 100.0;
    printf("%-46s core clock %7.0f MHz, %5.2f shader cycles per dependent v_add, %7.2f us\n", what, clk_mhz,
           static_cast<double>(h[0]) / (16.0 * n), h[1] / 100.0);
}

int main()
{
    float *d_out;
    unsigned long long *d_st, h[2];
    float4 *d_big;
    const size_t big = 300u << 20;
    hipMalloc(&d_out, 1 << 20);
    hipMalloc(&d_st, 64);
    hipMalloc(&d_big, big);
    hipMemset(d_big, 0, big);
    const int n = 600; // 9600 dependent adds: ~20-40 us
    for (int blocks : {1, 190}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            usleep(20000);
            hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(blocks == 1 ? 64 : 256), 0, 0, d_out, d_st, n, 1.0f);
            hipMemcpy(h, d_st, 16, hipMemcpyDeviceToHost);
            char what[96];
            snprintf(what, sizeof what, "%d block(s), after 20 ms idle", blocks);
            report(what, h, n);
        }
        for (int i = 0; i < 200; ++i)
            hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(blocks == 1 ? 64 : 256), 0, 0, d_out, d_st, n, 1.0f);
        hipMemcpy(h, d_st, 16, hipMemcpyDeviceToHost);
        char what[96];
        snprintf(what, sizeof what, "%d block(s), 200 launches back to back", blocks);
        report(what, h, n);
        for (int i = 0; i < 50; ++i) {
            hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, 0, d_big, d_out, big / 16);
            hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(blocks == 1 ? 64 : 256), 0, 0, d_out, d_st, n, 1.0f);
        }
        hipMemcpy(h, d_st, 16, hipMemcpyDeviceToHost);
        snprintf(what, sizeof what, "%d block(s), behind a 300 MB streaming kernel", blocks);
        report(what, h, n);
    }
    return 0;
}
