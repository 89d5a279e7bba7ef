// does v_mfma_f32_16x16x32_f16 keep binary16 subnormal INPUTS?  and how accurate is its f32 accumulation?
// hipcc --offload-arch=gfx950 -O2 -o denorm_probe denorm_probe.hip && ./denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave: D = A(16 x K) * B(K x 16), rows given row-major as f16 (a[m][k], b[n][k])
__global__ void k_mfma(const _Float16 *a, const _Float16 *b, int K, float *d)
{
    const int lane = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 32) {
        half8 fa = *reinterpret_cast<const half8 *>(a + (lane & 15) * K + k0 + (lane >> 4) * 8);
        half8 fb = *reinterpret_cast<const half8 *>(b + (lane & 15) * K + k0 + (lane >> 4) * 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i)
        d[((lane >> 4) * 4 + i) * 16 + (lane & 15)] = acc[i];
}

int main()
{
    const int K = 1024;
    std::vector<_Float16> a(16 * K), b(16 * K);
    // test 1: subnormal inputs.  a[0][:] = 2^-20 (subnormal: < 2^-14), b[0][:] = 1024  -> exact dot = K * 2^-10 = 1.0
    for (int k = 0; k < K; ++k) {
        for (int m = 0; m < 16; ++m) {
            a[m * K + k] = (_Float16)0.0f;
            b[m * K + k] = (_Float16)0.0f;
        }
        a[0 * K + k] = (_Float16)9.5367431640625e-07f;  // 2^-20
        b[0 * K + k] = (_Float16)1024.0f;
        a[1 * K + k] = (_Float16)9.5367431640625e-07f;  // subnormal x subnormal-ish: b[1] = 2^-15 (subnormal) -> product 2^-35
        b[1 * K + k] = (_Float16)3.0517578125e-05f;     // 2^-15
    }
    _Float16 *da, *db;
    float *dd;
    hipMalloc(&da, a.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&dd, 256 * 4);
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(da, db, K, dd);
    std::vector<float> d(256);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    printf("subnormal a x 1024        : got %.9g want 1 (0 = inputs flushed)\n", d[0 * 16 + 0]);
    printf("subnormal a x subnormal b : got %.9g want %.9g\n", d[1 * 16 + 1], K * std::ldexp(1.0, -35));
    // test 2: accumulation accuracy on random unit-ish vectors (values are exact f16, so the only error is the accumulate)
    srand(7);
    double worst_rel = 0, worst_vs_seq = 0;
    for (int rep = 0; rep < 200; ++rep) {
        for (int i = 0; i < 16 * K; ++i) {
            a[i] = (_Float16)(((rand() % 2001) - 1000) / 1000.0f / 32.0f);
            b[i] = (_Float16)(((rand() % 2001) - 1000) / 1000.0f / 32.0f);
        }
        hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
        k_mfma<<<1, 64>>>(da, db, K, dd);
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ex = 0, ab = 0;
                float seq = 0.0f;
                for (int k = 0; k < K; ++k) {
                    const float p = (float)a[m * K + k] * (float)b[n * K + k];
                    ex += (double)p;
                    ab += std::fabs((double)p);
                    seq = seq + p;
                }
                worst_rel = std::fmax(worst_rel, std::fabs(d[m * 16 + n] - ex) / ab);
                worst_vs_seq = std::fmax(worst_vs_seq, std::fabs((double)d[m * 16 + n] - (double)seq) / ab);
            }
    }
    printf("accumulation: max |mfma - exact| / sum|p| = %.3g  (K * 2^-24 = %.3g; 2^-24 = %.3g)\n", worst_rel, K * std::ldexp(1.0, -24), std::ldexp(1.0, -24));
    printf("              max |mfma - sequential f32| / sum|p| = %.3g\n", worst_vs_seq);
    return 0;
}
