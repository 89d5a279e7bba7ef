// Is a K = 1 f32 matrix-core instruction one correctly rounded step fl(c + a*b) per element -- i.e. can a chain of them
// reproduce the reference's strict left-to-right dot product (rag_engine.rs:1777-1779) bit for bit when a*b is exact in f32
// (binary16 rows widened to f32: 11-bit x 11-bit significands)?  Also maps the accumulator layout of the 2-block form.
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o mfma_f32_chain_probe mfma_f32_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float v32f __attribute__((ext_vector_type(32)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// a, b: [2 blocks][32 rows][K] f32 (values exactly representable in binary16)
__global__ void k_32x32x1(const float *a, const float *b, int K, float *d)
{
    const int lane = threadIdx.x;
    const float *ra = a + (size_t)lane * K, *rb = b + (size_t)lane * K; // lane = block * 32 + row
    v32f acc;
    for (int i = 0; i < 32; ++i) acc[i] = 0.0f;
    for (int k = 0; k < K; ++k)
        acc = __builtin_amdgcn_mfma_f32_32x32x1f32(ra[k], rb[k], acc, 0, 0, 0);
    for (int v = 0; v < 32; ++v)
        d[v * 64 + lane] = acc[v];
}
// one block, K = 2 per instruction: lanes 0-31 supply k, lanes 32-63 supply k + 1
__global__ void k_32x32x2(const float *a, const float *b, int K, float *d)
{
    const int lane = threadIdx.x;
    const float *ra = a + (size_t)(lane & 31) * K + (lane >> 5), *rb = b + (size_t)(lane & 31) * K + (lane >> 5);
    v16f acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (int k = 0; k < K; k += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[k], rb[k], acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v)
        d[v * 64 + lane] = acc[v];
}
// 16x16x4: lane l supplies row l & 15 at k + (l >> 4)
__global__ void k_16x16x4(const float *a, const float *b, int K, float *d)
{
    const int lane = threadIdx.x;
    const float *ra = a + (size_t)(lane & 15) * K + (lane >> 4), *rb = b + (size_t)(lane & 15) * K + (lane >> 4);
    v4f acc = {0, 0, 0, 0};
    for (int k = 0; k < K; k += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[k], rb[k], acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v)
        d[v * 64 + lane] = acc[v];
}

static float seq_dot(const float *x, const float *y, int K)
{
    float s = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float p = x[k] * y[k];
        s = s + p;
    }
    return s;
}

int main()
{
    const int K = 1024;
    std::vector<float> a(64 * K), b(64 * K), d(32 * 64);
    float *da, *db, *dd;
    hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dd, d.size() * 4);
    srand(11);
    long bad1 = 0, bad2 = 0, bad4 = 0, n1 = 0, n2 = 0, n4 = 0;
    bool layout_ok = true;
    for (int rep = 0; rep < 60; ++rep) {
        // binary16-representable values of mixed magnitude (cancellation, tiny terms, a few subnormal binary16 values)
        for (size_t i = 0; i < a.size(); ++i) {
            const int e1 = rep % 3 == 0 ? 0 : -(rand() % 12), e2 = rep % 3 == 0 ? 0 : -(rand() % 12);
            a[i] = (float)(_Float16)(std::ldexp(((rand() % 2049) - 1024) / 1024.0f, e1 - 3));
            b[i] = (float)(_Float16)(std::ldexp(((rand() % 2049) - 1024) / 1024.0f, e2 - 3));
            if (rand() % 97 == 0) a[i] = (float)(_Float16)std::ldexp((float)(rand() % 1024), -24); // binary16 subnormal
        }
        hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
        k_32x32x1<<<1, 64>>>(da, db, K, dd);
        hipMemcpy(d.data(), dd, 32 * 64 * 4, hipMemcpyDeviceToHost);
        for (int v = 0; v < 32; ++v)
            for (int lane = 0; lane < 64; ++lane) {
                // guessed layout: block = v / 16; i = 8 * ((v % 16) / 4) + 4 * (lane / 32) + v % 4; j = lane % 32
                const int blk = v / 16, i = 8 * ((v % 16) / 4) + 4 * (lane / 32) + v % 4, j = lane % 32;
                const float want = seq_dot(&a[(size_t)(blk * 32 + i) * K], &b[(size_t)(blk * 32 + j) * K], K);
                ++n1;
                if (memcmp(&want, &d[v * 64 + lane], 4) != 0) {
                    ++bad1;
                    if (rep == 0 && bad1 < 4) { layout_ok = false; printf("  32x32x1: v %d lane %d got %a want %a\n", v, lane, d[v * 64 + lane], want); }
                }
            }
        k_32x32x2<<<1, 64>>>(da, db, K, dd);
        hipMemcpy(d.data(), dd, 16 * 64 * 4, hipMemcpyDeviceToHost);
        for (int v = 0; v < 16; ++v)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = 8 * (v / 4) + 4 * (lane / 32) + v % 4, j = lane % 32;
                const float want = seq_dot(&a[(size_t)i * K], &b[(size_t)j * K], K);
                ++n2;
                bad2 += memcmp(&want, &d[v * 64 + lane], 4) != 0;
            }
        k_16x16x4<<<1, 64>>>(da, db, K, dd);
        hipMemcpy(d.data(), dd, 4 * 64 * 4, hipMemcpyDeviceToHost);
        for (int v = 0; v < 4; ++v)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = 4 * (lane / 16) + v, j = lane % 16;
                const float want = seq_dot(&a[(size_t)i * K], &b[(size_t)j * K], K);
                ++n4;
                bad4 += memcmp(&want, &d[v * 64 + lane], 4) != 0;
            }
    }
    printf("v_mfma_f32_32x32x1_2b_f32 chain vs strict sequential f32 (exact products): %ld of %ld differ (layout guess %s)\n", bad1, n1, layout_ok ? "ok" : "WRONG?");
    printf("v_mfma_f32_32x32x2_f32    chain vs strict sequential f32                  : %ld of %ld differ\n", bad2, n2);
    printf("v_mfma_f32_16x16x4_f32    chain vs strict sequential f32                  : %ld of %ld differ\n", bad4, n4);
    return 0;
}
