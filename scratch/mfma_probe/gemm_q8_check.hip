// Direct check of launch_gemm_q8 (materialise mode) against the integer arithmetic on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../../rust-local-rag_amd/csrc/kernels.h"
using namespace rlr;
int main()
{
    const uint32_t n = 700, dim = 768, nq = 40;
    std::vector<float> rows(n * dim), qs(nq * dim);
    srand(3);
    for (auto &v : rows) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    for (auto &v : qs) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    float *d_rows, *d_scale, *d_scores, *d_qscale; void *d_q8, *d_img, *d_qfrag; uint32_t *d_stats;
    hipMalloc(&d_rows, rows.size() * 4); hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&d_q8, n * dim); hipMalloc(&d_scale, n * 4); hipMalloc(&d_stats, 16); hipMemset(d_stats, 0, 16);
    launch_q8_build(d_rows, dim * 4 / 16, dim, 0, n, d_q8, d_scale, d_stats, nullptr);
    hipMalloc(&d_img, q8_image_bytes(dim, n));
    launch_q8_image_build(d_q8, dim, n, 0, (n + 255) / 256, d_img, nullptr);
    std::vector<int8_t> codes(nq * dim), frag(q8_query_frag_bytes(nq, dim));
    std::vector<float> qscale(nq);
    for (uint32_t q = 0; q < nq; ++q) {
        float m = 0; for (uint32_t i = 0; i < dim; ++i) m = std::max(m, std::fabs(qs[q * dim + i]));
        const float t = m / 127.0f; qscale[q] = t;
        for (uint32_t i = 0; i < dim; ++i) codes[q * dim + i] = (int8_t)std::nearbyint(qs[q * dim + i] / t);
    }
    q8_pack_queries(codes.data(), nq, dim, frag.data());
    hipMalloc(&d_qfrag, frag.size()); hipMemcpy(d_qfrag, frag.data(), frag.size(), hipMemcpyHostToDevice);
    hipMalloc(&d_qscale, nq * 4); hipMemcpy(d_qscale, qscale.data(), nq * 4, hipMemcpyHostToDevice);
    const size_t stride = 704;
    hipMalloc(&d_scores, nq * stride * 4); hipMemset(d_scores, 0xFF, nq * stride * 4);
    hipError_t e = launch_gemm_q8(d_img, d_scale, dim, 0, n, d_qfrag, d_qscale, nq, nullptr, nullptr, 0, nullptr, d_scores, stride, nullptr);
    hipDeviceSynchronize();
    printf("launch: %s / %s\n", hipGetErrorString(e), hipGetErrorString(hipGetLastError()));
    std::vector<float> sc(nq * stride), scale(n); std::vector<uint8_t> q8(n * dim);
    hipMemcpy(sc.data(), d_scores, sc.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(scale.data(), d_scale, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(q8.data(), d_q8, n * dim, hipMemcpyDeviceToHost);
    int bad = 0; double worst = 0;
    for (uint32_t q = 0; q < nq; ++q) for (uint32_t r = 0; r < n; ++r) {
        long acc = 0; for (uint32_t i = 0; i < dim; ++i) acc += (long)((int)q8[r * dim + i] - 128) * codes[q * dim + i];
        const float want = scale[r] * (qscale[q] * (float)acc);
        const float got = sc[q * stride + r];
        if (!(got == want)) { if (bad < 5) printf("q %u r %u got %g want %g\n", q, r, got, want); ++bad; }
        double ex = 0; for (uint32_t i = 0; i < dim; ++i) ex += (double)rows[r * dim + i] * qs[q * dim + i];
        worst = std::max(worst, std::fabs(ex - want));
    }
    printf("mismatches %d of %u, worst |exact - nominated| = %g\n", bad, nq * n, worst);
    return bad != 0;
}
