// TEST INFRASTRUCTURE -- a CPU stand-in for the HIP runtime entry points librlr_gpu.so imports, so that the HOST half of
// the HIP translation units (csrc/index.hip, lexical.hip, exact.hip, gemm.hip, ...: context and workspace pools, the
// readers/writer lock, condition variables, leases, tickets, error paths) can run under ThreadSanitizer on the CPU build
// (GPU sanitizers are not available on this pool).  "Device" memory is host memory, copies and fills run at once on the
// calling thread (so two host threads sharing one workspace show up as a data race on its bytes), kernel launches do
// nothing -- results are meaningless, only the synchronisation of the host code is under test.
// Nothing here ships: it is linked only into tests/sanitize/tsan_stress (see tests/test_host_sanitize_cpu.py).
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <cstdlib>
#include <cstring>

namespace {
struct StubStream {
    // deliberately NOT atomic: a stream belongs to one leased context / workspace at a time, so two host threads
    // enqueuing on one stream without a hand-over through the pool's mutex are reported by ThreadSanitizer
    unsigned long ops = 0;
};
struct StubEvent {
    std::atomic<unsigned long> stamp{0};
};
thread_local int t_device = 0;
thread_local struct {
    dim3 grid, block;
    size_t shmem;
    hipStream_t stream;
} t_cfg;
void touch(hipStream_t s)
{
    if (s)
        reinterpret_cast<StubStream *>(s)->ops += 1;
}
} // namespace

extern "C" {

void **__hipRegisterFatBinary(const void *)
{
    static void *handle = nullptr;
    return &handle;
}
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned int, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream)
{
    t_cfg.grid = grid;
    t_cfg.block = block;
    t_cfg.shmem = shmem;
    t_cfg.stream = stream;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream)
{
    *grid = t_cfg.grid;
    *block = t_cfg.block;
    *shmem = t_cfg.shmem;
    *stream = t_cfg.stream;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void *, dim3, dim3, void **, size_t, hipStream_t s)
{
    touch(s);
    return hipSuccess;
}

hipError_t hipGetDeviceCount(int *n)
{
    *n = 1;
    return hipSuccess;
}
hipError_t hipSetDevice(int d)
{
    if (d != 0)
        return hipErrorInvalidDevice;
    t_device = d;
    return hipSuccess;
}
hipError_t hipGetDevice(int *d)
{
    *d = t_device;
    return hipSuccess;
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int)
{
    std::memset(p, 0, sizeof *p);
    std::strcpy(p->name, "stub");
    std::strcpy(p->gcnArchName, "gfx950");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = 64ull << 30;
    p->sharedMemPerBlock = 160 << 10;
    p->maxSharedMemoryPerMultiProcessor = 160 << 10;
    p->warpSize = 64;
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int)
{
    switch (a) {
    case hipDeviceAttributeMultiprocessorCount: *v = 256; break;
    case hipDeviceAttributeMaxSharedMemoryPerBlock: *v = 160 << 10; break;
    case hipDeviceAttributeWarpSize: *v = 64; break;
    default: *v = 0; break;
    }
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "stub error"; }
hipError_t hipGetLastError(void) { return hipSuccess; }

hipError_t hipMalloc(void **p, size_t n)
{
    *p = std::calloc(n ? n : 1, 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p)
{
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned int)
{
    *p = std::calloc(n ? n : 1, 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p)
{
    std::free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind)
{
    if (n)
        std::memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t st)
{
    touch(st);
    if (n)
        std::memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t st)
{
    touch(st);
    if (n)
        std::memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipDeviceCanAccessPeer(int *can, int, int)
{
    *can = 0;
    return hipSuccess;
}
hipError_t hipDeviceEnablePeerAccess(int, unsigned int) { return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t st)
{
    touch(st);
    for (size_t r = 0; r < h; ++r)
        std::memmove(static_cast<char *>(d) + r * dp, static_cast<const char *>(s) + r * sp, w);
    return hipSuccess;
}
hipError_t hipMemset(void *d, int v, size_t n)
{
    if (n)
        std::memset(d, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st)
{
    touch(st);
    if (n)
        std::memset(d, v, n);
    return hipSuccess;
}

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int)
{
    *s = reinterpret_cast<hipStream_t>(new StubStream());
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s)
{
    delete reinterpret_cast<StubStream *>(s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s)
{
    touch(s);
    // stand-in for the time the device takes: without it calls finish at once and the pools are never exhausted
    static const int us = getenv("STUB_SYNC_US") ? atoi(getenv("STUB_SYNC_US")) : 0;
    if (us > 0 && s)
        std::this_thread::sleep_for(std::chrono::microseconds(us));
    return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t s)
{
    touch(s);
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned int)
{
    touch(s);
    if (e)
        (void)reinterpret_cast<StubEvent *>(e)->stamp.load(std::memory_order_acquire);
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e)
{
    *e = reinterpret_cast<hipEvent_t>(new StubEvent());
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned int) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e)
{
    delete reinterpret_cast<StubEvent *>(e);
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    touch(s);
    reinterpret_cast<StubEvent *>(e)->stamp.fetch_add(1, std::memory_order_release);
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t)
{
    *ms = 0.0f;
    return hipSuccess;
}

} // extern "C"
