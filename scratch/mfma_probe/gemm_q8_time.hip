// Raw speed of the int8 nomination GEMM (filter mode, threshold +inf so nothing is appended): 10 M x 768, 256 queries.
#include "gemm_q8.hip"
#include <cstdio>
#include <vector>
using namespace rlr;
int main()
{
    const uint32_t n = 10'000'000, dim = 768, nq = 256;
    void *d_img, *d_qfrag; float *d_scale, *d_qscale, *d_tau; SelectState *d_st; uint64_t *d_cand;
    hipMalloc(&d_img, q8_image_bytes(dim, n)); hipMemset(d_img, 1, q8_image_bytes(dim, n));
    hipMalloc(&d_scale, n * 4); hipMemset(d_scale, 0, n * 4);
    hipMalloc(&d_qfrag, q8_query_frag_bytes(nq, dim)); hipMemset(d_qfrag, 1, q8_query_frag_bytes(nq, dim));
    hipMalloc(&d_qscale, nq * 4); hipMemset(d_qscale, 0, nq * 4);
    std::vector<float> tau(nq, INFINITY);
    hipMalloc(&d_tau, nq * 4); hipMemcpy(d_tau, tau.data(), nq * 4, hipMemcpyHostToDevice);
    hipMalloc(&d_st, nq * sizeof(SelectState)); hipMemset(d_st, 0, nq * sizeof(SelectState));
    hipMalloc(&d_cand, nq * 8192 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; ++it)
        launch_gemm_q8(d_img, d_scale, dim, 0, n, d_qfrag, d_qscale, nq, d_tau, d_cand, 8192, d_st, nullptr, 0, nullptr);
    hipEventRecord(a, nullptr);
    for (int it = 0; it < 10; ++it)
        launch_gemm_q8(d_img, d_scale, dim, 0, n, d_qfrag, d_qscale, nq, d_tau, d_cand, 8192, d_st, nullptr, 0, nullptr);
    hipEventRecord(b, nullptr); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("gemm_q8 10M x 768 x 256 queries: %.3f ms per pass = %.1f TOP/s, %.2f TB/s of codes\n", ms / 10,
           2.0 * nq * n * dim / (ms / 10 * 1e-3) / 1e12, (double)n * dim / (ms / 10 * 1e-3) / 1e12);
    return 0;
}
