// A/B harness for the Gram kernel: includes the library source so compiler flags / source variants can be compared in
// one GPU call.  usage: gram_bench <n_queries> <P> <dim> <f16:0|1> <n_rows> <reps>
#include "../../rust-local-rag_amd/csrc/exact.hip"
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
namespace rlr {
hipError_t dev_malloc(void **p, size_t bytes) { return hipMalloc(p, bytes); }
bool poison_mode() { return false; }
int32_t set_error(int32_t code, const char *, ...) { return code; }
}
__global__ void fill_kernel(uint32_t *p, size_t n_words, int f16)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n_words; i += gridDim.x * 256ull) {
        uint32_t h = static_cast<uint32_t>(i) * 2654435761u ^ static_cast<uint32_t>(i >> 32) * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        if (f16) // two binary16 of magnitude ~0.03, random signs
            p[i] = (0x2800u + (h & 0x3FFu)) | ((h >> 10 & 1u) << 15) | ((0x2800u + (h >> 11 & 0x3FFu)) << 16) | ((h >> 21 & 1u) << 31);
        else     // f32 in +-[2^-6, 2^-5)
            p[i] = 0x3C800000u | (h & 0x7FFFFFu) | ((h >> 23 & 1u) << 31);
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv)
{
    const uint32_t nq = argc > 1 ? atoi(argv[1]) : 1, P = argc > 2 ? atoi(argv[2]) : 308, dim = argc > 3 ? atoi(argv[3]) : 768;
    const int f16 = argc > 4 ? atoi(argv[4]) : 0;
    const uint32_t n_rows = argc > 5 ? atoi(argv[5]) : 100000, reps = argc > 6 ? atoi(argv[6]) : 200;
    const size_t esz = f16 ? 2 : 4;
    const uint32_t pitch16 = (dim * esz + 15) / 16;
    std::mt19937 rng(7);
    const size_t bytes = static_cast<size_t>(n_rows) * pitch16 * 16;
    std::vector<uint32_t> list(static_cast<size_t>(nq) * P);
    for (auto &v : list) v = rng() % n_rows;
    void *d_rows; uint32_t *d_list; float *d_gram;
    CK(hipMalloc(&d_rows, bytes)); CK(hipMalloc(&d_list, list.size() * 4)); CK(hipMalloc(&d_gram, static_cast<size_t>(nq) * P * P * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, static_cast<uint32_t *>(d_rows), bytes / 4, f16);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(d_list, list.data(), list.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(rlr::launch_gram_rows(d_rows, pitch16, dim, f16 ? RLR_F16 : RLR_F32, d_list, P, d_gram, nq, s));
    CK(hipStreamSynchronize(s));
    float best = 1e30f, sum = 0;
    for (uint32_t r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, s));
        CK(rlr::launch_gram_rows(d_rows, pitch16, dim, f16 ? RLR_F16 : RLR_F32, d_list, P, d_gram, nq, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    std::vector<float> g(static_cast<size_t>(P) * P);
    CK(hipMemcpy(g.data(), d_gram + static_cast<size_t>(nq - 1) * P * P, g.size() * 4, hipMemcpyDeviceToHost));
    uint64_t cs = 0;
    for (float v : g) { uint32_t b = __builtin_bit_cast(uint32_t, v); cs = cs * 1000003u + b; }
    printf("nq %u P %u dim %u f16 %d: avg %.1f us  min %.1f us  checksum %016llx\n", nq, P, dim, f16, sum / reps * 1e3, best * 1e3,
           static_cast<unsigned long long>(cs));
    return 0;
}
