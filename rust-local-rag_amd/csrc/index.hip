// index.hip -- the C ABI of include/rlr_gpu.h: device-resident chunk-embedding matrix,
// per-call search contexts (stream + workspace), and the kernel pipeline of a single query
//   scan (+digit-1 histogram) -> tail stage 1 -> tail stage 2 (tail.hip: select, reference-order re-score, sort, results
//   and a completion word straight into pinned host memory), or above 4 M rows the four specialised launches
//   hist2_find1 -> collect_find2 -> rescore_staged -> sort_emit; batches of queries go through the matrix cores (gemm.hip).
// There is no CPU compute path in this file: without a HIP device every compute entry
// point fails with RLR_E_NO_DEVICE.
#include "../../include/rlr_gpu.h"
#include "common.h"
#include "kernels.h"
#include "lds_select.h"
#include "sort_emit.h"
#include "pool_prepare.h"
#include "lexical_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <time.h>

using namespace rlr;

namespace {

thread_local char g_err[512] = "";

int32_t fail(int32_t code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

} // namespace

namespace rlr {
// the same thread-local message for the other translation units of the library (lexical.hip)
int32_t set_error(int32_t code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
} // namespace rlr

namespace {

#define RLR_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? RLR_E_OOM : RLR_E_HIP, "%s failed: %s (%s:%d)", \
                        #call, hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

#define RLR_TRY(call)                                                                              \
    do {                                                                                           \
        int32_t s_ = (call);                                                                       \
        if (s_ != RLR_OK)                                                                          \
            return s_;                                                                             \
    } while (0)

constexpr uint32_t kLdsSortCap = 4096; // candidates the single-workgroup sort can take
constexpr uint32_t kMaxDim = 8192;

uint32_t next_pow2(uint32_t v)
{
    uint32_t p = 1;
    while (p < v)
        p <<= 1;
    return p;
}

// The allocation behind an index's row matrix.  Where the rows land in physical memory moves the streaming rate of the scan by
// up to 1.5 %, so the policy is explicit: see rows_alloc.
struct RowBlock {
    void *raw = nullptr;     // what hipFree / hipMemAddressFree takes
    size_t raw_bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles; // virtual-memory form: one physical allocation per slab
    size_t slab = 0;
};

// how long the recent waits of one kind took (the hybrid wait sleeps through most of that before it polls)
struct WaitEma {
    double us = 0.0;
    uint32_t key = 0; // what "one kind" means to the caller (number of queries, ...): a change resets the average
};

// -------- per-call context ---------------------------------------------------------
struct Ctx {
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // device
    float *d_query = nullptr;      // q_cap x q_pitch floats
    uint32_t q_cap = 0;
    float *d_scores = nullptr;     // score_cap floats
    uint64_t score_cap = 0;
    uint32_t *d_hist = nullptr;    // 2 * kHistBins
    SelectState *d_state = nullptr; // q_cap states
    uint32_t *d_cand = nullptr;    // cand_cap
    uint64_t *d_packed = nullptr;  // cand_cap (power of two)
    uint32_t cand_cap = 0;
    uint64_t *d_out = nullptr;     // out_cap packed results
    uint64_t out_cap = 0;
    uint32_t *d_list = nullptr;    // row lists (score_rows / fetch / mmr)
    float *d_vals = nullptr;       // float outputs for lists / mmr
    uint32_t list_cap = 0;
    float *d_pool = nullptr;       // mmr pool P x dim, then gram P x P
    uint64_t pool_cap = 0;         // floats
    // batched (MFMA) path workspace
    void *d_qfrag = nullptr;        // binary16 fragment-major queries
    uint64_t qfrag_cap = 0;         // bytes
    float *d_tau = nullptr;         // per-query nomination threshold
    SelectState *d_bstate = nullptr;
    uint32_t *d_bhist = nullptr;    // q x 2 x kHistBins
    uint32_t *d_bstatus = nullptr;
    uint32_t *d_gsync = nullptr;   // 256 words: sibling-group arrival counters of the persistent batched GEMM (gemm.hip)
    void *h_batch = nullptr;       // pinned: SelectState[bq_cap] | status[bq_cap] (pageable staging makes the async copies synchronous)
    uint32_t bq_cap = 0;            // queries the four arrays above are sized for
    uint64_t *d_bcand = nullptr;    // q x fin_cap packed candidates
    uint64_t bcand_cap = 0;         // entries
    float *d_sample = nullptr;      // q x S nominated scores of the sample rows
    uint64_t sample_cap = 0;        // floats
    hipEvent_t bev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    // pinned host
    void *h_pin = nullptr;
    size_t h_pin_bytes = 0;
    std::vector<float> q_norm; // ||q||_2 of the staged queries (band of the 8-bit nomination scan)
    WaitEma wait_ema;          // how long the last waits for a call's completion words took (wait_flags)
    const float *h_q_kq = nullptr; // set for the duration of a call whose scans take their query in the kernel arguments
                                   // (ScanArgs::query_host): the staged queries in pinned host memory, q_pitch floats each
    bool hist_dirty = false; // a pipeline was enqueued and did not complete: d_hist may hold counts
    uint32_t *h_assert = nullptr; // RLR_POISON_ALLOC=1 only: pinned word the zero-histogram assertion kernel counts into
    // rlr_search_topk_device_begin / _end
    hipStream_t pending_stream = nullptr;
    uint32_t pending_q = 0, pending_k = 0;
    bool pending_timed = false;
    const uint64_t *pending_meta = nullptr;
};

} // namespace

struct rlr_index {
    uint32_t dim = 0;
    int32_t dtype = RLR_F32;
    int32_t device = 0;
    uint32_t pitch16 = 0;   // row pitch in 16-byte units
    uint32_t q_pitch = 0;   // floats per staged query (row pitch in elements)
    uint64_t n_rows = 0;
    uint64_t cap_rows = 0;
    void *d_rows = nullptr;
    RowBlock rows_block;      // how d_rows was obtained (rows_alloc / rows_free)
    int n_cu = 256;
    int scan_variant = 0;
    int fused_tail = -1;         // select -> re-score -> sort behind the scan in two launches (tail.hip): RLR_TAIL=1 always, 0 never
                                 // (the five-launch form), unset: while the corpus is small enough for the one-pass mode to be the rule
    uint32_t tail_direct_max = 3584; // RLR_TAIL_DIRECT_MAX: most scores in/above the k-th score's digit-1 bin for the one-pass mode (0: always
                                     // refine).  Re-scoring a few thousand candidates costs less than the two extra passes of the refine
                                     // mode (they spread over the workgroups that found them); 512 slots of the 4096 stay for the guard band.
                                     // A call that builds an MMR pool from the candidates caps it at 1024 (what the pool kernel takes).
    // Hybrid (text) searches: how many rows by cosine the blend asks the tail for beyond the `need` it keeps.  A lexical term
    // only ever ADDS to a row's blended score (weights and BM25 scores are non-negative as a rule), so the need + 8 best
    // cosines already hold every non-lexical row that can reach the pool, whatever the lexical rows among them do; the
    // blend kernel checks that on what it got -- its need-th blended score must beat anything an unfetched row can reach,
    // status 2 and the host's widening path else (a negative lexical weight, a tie across the boundary).  Rounds 2-3
    // fetched need + n_lexical + 8 (every lexical row counted as if it displaced one): 1808 rows instead of 332 for a
    // top-100 text search, a crowded digit-1 bin and ~25 us more tail.  RLR_HYBRID_FETCH=full at creation restores that.
    bool hybrid_fetch_full = false;
    uint32_t batch_min = 0;   // smallest batch that takes the matrix-core path; 0 = decide by the cost model,
                              // RLR_BATCH_MIN=n forces a threshold (a huge n disables the path)
    float max_row_sumsq = 1.0f; // largest sum of squares of a row stored with normalize_on_device = 0 (>= 1): the guard
                                // bands are derived for unit-norm operands and scale with |row| * |query|
    bool image_enabled = false; // keep a binary16 nomination image of the rows for the batched GEMM
    bool image_scan = false;    // single queries nominate over the image too (half the bytes of f32 rows)
    // optional 8-bit nomination copy for single queries (q8.hip): a quarter of the f32 bytes
    bool q8_enabled = false;
    void *d_q8 = nullptr;        // cap_rows x dim bytes
    float *d_q8_scale = nullptr; // cap_rows
    uint32_t *d_q8_stats = nullptr; // [0] max row error norm (float bits), [1] max scale (float bits), [2] an Inf row exists
    uint64_t q8_cap_rows = 0;
    float q8_delta = 0.0f, q8_scale_max = 0.0f;
    bool q8_has_inf = false;
    void *d_image = nullptr;
    size_t image_cap = 0;       // bytes
    std::mutex mu;
    std::condition_variable ctx_cv; // callers beyond ctx_cap wait here for a context to come back
    std::vector<Ctx *> free_ctx;
    int ctx_made = 0;               // contexts alive (in free_ctx or leased)
    int ctx_cap = 16;               // RLR_MAX_CONTEXTS (1..64)
    bool profiling = false;
    rlr_profile prof{};
};

namespace {

size_t row_bytes(const rlr_index *ix)
{
    return static_cast<size_t>(ix->pitch16) * 16;
}

} // namespace

namespace rlr {
bool poison_mode()
{
    static const bool poison = [] {
        const char *v = getenv("RLR_POISON_ALLOC");
        return v && v[0] == '1';
    }();
    return poison;
}

hipError_t dev_malloc(void **p, size_t bytes)
{
    const bool poison = poison_mode();
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess)
        (void)hipGetLastError(); // reported through the return value; do not leave it for a later launch check to find
    if (e == hipSuccess && poison && bytes) {
        e = hipMemset(*p, 0xFF, bytes);
        if (e == hipSuccess)
            e = hipDeviceSynchronize();
    }
    return e;
}
} // namespace rlr

namespace {
int32_t use_device(const rlr_index *ix)
{
    RLR_HIP(hipSetDevice(ix->device));
    return RLR_OK;
}

template <typename T>
int32_t grow(T **p, uint64_t *cap, uint64_t want, bool keep = false)
{
    if (*cap >= want && *p)
        return RLR_OK;
    T *n = nullptr;
    RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&n), want * sizeof(T)));
    if (keep && *p && *cap) {
        RLR_HIP(hipMemcpy(n, *p, *cap * sizeof(T), hipMemcpyDeviceToDevice));
        RLR_HIP(hipStreamSynchronize(nullptr)); // a device-to-device copy may return before it has run
    }
    if (*p)
        (void)hipFree(*p);
    *p = n;
    *cap = want;
    return RLR_OK;
}

int32_t pin_reserve(Ctx *c, size_t bytes)
{
    if (c->h_pin_bytes >= bytes)
        return RLR_OK;
    if (c->h_pin)
        (void)hipHostFree(c->h_pin);
    c->h_pin = nullptr;
    c->h_pin_bytes = 0;
    size_t want = std::max<size_t>(bytes, 1 << 16);
    RLR_HIP(hipHostMalloc(&c->h_pin, want, hipHostMallocDefault));
    c->h_pin_bytes = want;
    return RLR_OK;
}

void ctx_free(Ctx *c)
{
    if (!c)
        return;
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    (void)hipFree(c->d_query);
    (void)hipFree(c->d_scores);
    (void)hipFree(c->d_hist);
    (void)hipFree(c->d_state);
    (void)hipFree(c->d_cand);
    (void)hipFree(c->d_packed);
    (void)hipFree(c->d_out);
    (void)hipFree(c->d_list);
    (void)hipFree(c->d_vals);
    (void)hipFree(c->d_pool);
    (void)hipFree(c->d_qfrag);
    (void)hipFree(c->d_tau);
    (void)hipFree(c->d_bstate);
    (void)hipFree(c->d_bhist);
    (void)hipFree(c->d_bstatus);
    (void)hipFree(c->d_gsync);
    (void)hipFree(c->d_bcand);
    (void)hipFree(c->d_sample);
    for (auto &e : c->bev)
        if (e) (void)hipEventDestroy(e);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_batch) (void)hipHostFree(c->h_batch);
    if (c->h_assert) (void)hipHostFree(c->h_assert);
    delete c;
}

// A free context, a new one while fewer than ctx_cap exist, else wait until a call hands one back.  The reference serves
// searches from one tokio worker per core under a read lock (src/main.rs:140, src/mcp_server.rs:89, :377): callers may be
// many, but every context costs a stream, pinned memory and n_rows x 4 B of scores, and more streams than hardware
// queues buy no overlap -- so the pool is bounded like the lexical index' workspaces and the surplus callers queue.
int32_t ctx_acquire(rlr_index *ix, Ctx **out)
{
    {
        std::unique_lock<std::mutex> lk(ix->mu);
        for (;;) {
            if (!ix->free_ctx.empty()) {
                static const bool rotate = getenv("RLR_CTX_ROTATE") != nullptr; // (experiment: cycle through the contexts)
                Ctx *c = rotate ? ix->free_ctx.front() : ix->free_ctx.back();
                if (rotate)
                    ix->free_ctx.erase(ix->free_ctx.begin());
                else
                    ix->free_ctx.pop_back();
                lk.unlock();
                if (c->hist_dirty) { // a previous call failed half way: restore the zero-histogram invariant
                    (void)hipStreamSynchronize(c->stream);
                    // (on the context's own stream: the null stream does not order against a non-blocking one)
                    if (hipMemsetAsync(c->d_hist, 0, 2 * kHistBins * sizeof(uint32_t), c->stream) == hipSuccess &&
                        (!c->d_state || hipMemsetAsync(c->d_state, 0, static_cast<size_t>(c->q_cap) * sizeof(SelectState), c->stream) ==
                                            hipSuccess))
                        c->hist_dirty = false;
                }
                *out = c;
                return RLR_OK;
            }
            if (ix->ctx_made < ix->ctx_cap) {
                ix->ctx_made++;
                break;
            }
            ix->ctx_cv.wait(lk);
        }
    }
    auto give_up = [ix](Ctx *c) {
        ctx_free(c);
        {
            std::lock_guard<std::mutex> lk(ix->mu);
            ix->ctx_made--;
        }
        ix->ctx_cv.notify_one();
    };
    Ctx *c = new (std::nothrow) Ctx();
    if (!c) {
        give_up(nullptr);
        return fail(RLR_E_OOM, "host allocation failed");
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; ++i)
        e = hipEventCreate(&c->ev[i]);
    for (int i = 0; i < 5 && e == hipSuccess; ++i)
        e = hipEventCreate(&c->bev[i]);
    if (e == hipSuccess)
        e = rlr::dev_malloc(reinterpret_cast<void **>(&c->d_hist), 2 * kHistBins * sizeof(uint32_t));
    // The zero-histogram invariant is established ON THE CONTEXT'S STREAM: a null-stream hipMemset of device
    // memory may return before it has run, and a non-blocking stream is not ordered against the null stream --
    // the first scan's histogram atomics could then land before the fill and be wiped (a rare wrong threshold
    // on the first search of a fresh context, caught by the multi-shard fuzz with five contexts starting at once).
    if (e == hipSuccess)
        e = hipMemsetAsync(c->d_hist, 0, 2 * kHistBins * sizeof(uint32_t), c->stream);
    if (e == hipSuccess && rlr::poison_mode()) {
        e = hipHostMalloc(reinterpret_cast<void **>(&c->h_assert), 64, hipHostMallocDefault);
        if (e == hipSuccess)
            *c->h_assert = 0;
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        give_up(c);
        return fail(RLR_E_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return RLR_OK;
}

void ctx_release(rlr_index *ix, Ctx *c)
{
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        ix->free_ctx.push_back(c);
    }
    ix->ctx_cv.notify_one();
}

struct CtxLease {
    rlr_index *ix;
    Ctx *c = nullptr;
    explicit CtxLease(rlr_index *i) : ix(i) {}
    ~CtxLease()
    {
        if (c)
            ctx_release(ix, c);
    }
};

int32_t check_handle(const rlr_index *ix)
{
    if (!ix)
        return fail(RLR_E_INVALID, "null index handle");
    return RLR_OK;
}

// RLR_ROWS_ALLOC (experiment switch; the default is chosen in DESIGN.md section 5):
//   "plain"          hipMalloc(bytes)
//   "align:<MiB>"    hipMalloc(bytes + A), base rounded up to A
//   "round:<MiB>"    hipMalloc(bytes rounded up to a multiple of A)
//   "vmm:<MiB>"      one virtual range (hipMemAddressReserve, aligned to the slab size) backed by separate physical
//                    allocations of <MiB> each (hipMemCreate + hipMemMap)
void rows_free(RowBlock *b)
{
    if (!b->handles.empty()) {
        (void)hipMemUnmap(b->raw, b->raw_bytes);
        for (auto h : b->handles)
            (void)hipMemRelease(h);
        (void)hipMemAddressFree(b->raw, b->raw_bytes);
    } else if (b->raw) {
        (void)hipFree(b->raw);
    }
    *b = RowBlock();
}

// Time of the scan kernel this index's searches launch over the rows at `base` (a slab of a fresh allocation: the
// content is irrelevant), best of `reps` launches after one warm-up, in ms; < 0 on error.
float slab_scan_ms(const rlr_index *ix, const void *base, size_t bytes, float *d_scratch, hipEvent_t ev0, hipEvent_t ev1, int reps)
{
    ScanArgs sa;
    sa.rows = base;
    sa.n_rows = static_cast<uint32_t>(bytes / row_bytes(ix));
    sa.scores = d_scratch;
    sa.query = d_scratch + sa.n_rows; // zeros
    sa.hist = nullptr;
    sa.dim = ix->dim;
    sa.pitch16 = ix->pitch16;
    sa.dtype = ix->dtype;
    sa.n_cu = ix->n_cu;
    sa.variant = ix->scan_variant;
    float best = -1.0f;
    for (int i = 0; i <= reps; ++i) {
        if (hipEventRecord(ev0, nullptr) != hipSuccess || launch_scan(sa, nullptr) != hipSuccess ||
            hipEventRecord(ev1, nullptr) != hipSuccess || hipEventSynchronize(ev1) != hipSuccess)
            return -1.0f;
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev0, ev1) != hipSuccess)
            return -1.0f;
        if (i > 0 && (best < 0 || ms < best))
            best = ms;
    }
    return best;
}

// "select" (an experiment that did NOT become the default): the row matrix on 1 GiB physical slabs behind one virtual range,
// the slabs chosen by measurement.  In the steady state the scan kernel streams a 1 GiB slab at 0.897-0.902 of the HBM peak or
// at 0.883-0.890, the same slabs every time, in runs of consecutive allocations (scratch/slab_scan.py, slab_content.py) -- so:
// map the slabs, time the scan kernel over each, swap every slab more than 0.7 % slower than the best for a fresh allocation
// (holding on to the rejects so the driver cannot hand the same memory back).  What it buys: nothing reliable.  During the
// first seconds after the mapping a slab's rate moves between the two grades from one measurement to the next, so the
// selection sorts noise; and a whole 10 M-row scan runs ~2 % below the mean of its slabs whatever they are (0.867 over slabs
// that average 0.892).  Six selected indexes in one process: 0.872 .. 0.883; six from hipMalloc: 0.869 .. 0.887
// (scratch/alloc_spread.py).  Kept behind RLR_ROWS_ALLOC=select for whoever wants to look again; DESIGN.md section 5.
hipError_t rows_alloc_select(const rlr_index *ix, size_t bytes, RowBlock *b, void **base)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ix->device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess || gran == 0)
        return e == hipSuccess ? hipErrorNotSupported : e;
    const size_t slab = ((1ull << 30) + gran - 1) / gran * gran;
    const size_t n_slabs = (bytes + slab - 1) / slab;
    const size_t total = n_slabs * slab;
    *b = RowBlock();
    e = hipMemAddressReserve(&b->raw, total, slab, nullptr, 0);
    if (e != hipSuccess)
        return e;
    b->raw_bytes = total;
    b->slab = slab;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    auto place = [&](size_t pos, hipMemGenericAllocationHandle_t h) -> hipError_t {
        char *at = static_cast<char *>(b->raw) + pos * slab;
        hipError_t pe = hipMemMap(at, slab, 0, h, 0);
        if (pe == hipSuccess)
            pe = hipMemSetAccess(at, slab, &acc, 1);
        return pe;
    };
    size_t mapped = 0;
    for (; mapped < n_slabs && e == hipSuccess; ++mapped) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, slab, &prop, 0);
        if (e != hipSuccess)
            break;
        b->handles.push_back(h);
        e = place(mapped, h);
    }
    if (e != hipSuccess) {
        for (size_t i = 0; i < b->handles.size(); ++i) {
            if (i < mapped)
                (void)hipMemUnmap(static_cast<char *>(b->raw) + i * slab, slab);
            (void)hipMemRelease(b->handles[i]);
        }
        (void)hipMemAddressFree(b->raw, total);
        *b = RowBlock();
        (void)hipGetLastError();
        return e;
    }
    *base = b->raw;
    // ---- selection (best effort: any failure from here on keeps the slabs as they are) ----
    static const bool no_select = getenv("RLR_ROWS_NO_SELECT") != nullptr;
    const size_t slab_rows = slab / row_bytes(ix);
    float *d_scratch = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipMemGenericAllocationHandle_t> rejects;
    if (!no_select && n_slabs >= 2 && slab_rows >= 4096 &&
        rlr::dev_malloc(reinterpret_cast<void **>(&d_scratch), (slab_rows + ix->q_pitch + 64) * sizeof(float)) == hipSuccess &&
        hipMemset(d_scratch, 0, (slab_rows + ix->q_pitch + 64) * sizeof(float)) == hipSuccess &&
        hipStreamSynchronize(nullptr) == hipSuccess && hipEventCreate(&ev0) == hipSuccess && hipEventCreate(&ev1) == hipSuccess) {
        std::vector<float> ms(n_slabs, -1.0f);
        float best = -1.0f;
        bool ok = true;
        for (size_t i = 0; i < n_slabs && ok; ++i) {
            ms[i] = slab_scan_ms(ix, static_cast<char *>(b->raw) + i * slab, slab, d_scratch, ev0, ev1, 3);
            ok = ms[i] > 0;
            if (ok && (best < 0 || ms[i] < best))
                best = ms[i];
        }
        size_t budget = std::min<size_t>(2 * n_slabs, 96); // fresh slabs to try in all
        uint32_t swapped = 0, tried = 0;
        for (size_t i = 0; i < n_slabs && ok && budget > 0; ++i) {
            for (int attempt = 0; attempt < 4 && ms[i] > best * 1.007f && budget > 0; ++attempt) {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 3 * slab)
                    budget = 1; // (this is the last try)
                hipMemGenericAllocationHandle_t h;
                if (hipMemCreate(&h, slab, &prop, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    budget = 0;
                    break;
                }
                budget--;
                tried++;
                char *at = static_cast<char *>(b->raw) + i * slab;
                if (hipMemUnmap(at, slab) != hipSuccess || place(i, h) != hipSuccess) {
                    // (cannot happen on a healthy runtime; the position must not stay unmapped)
                    (void)hipGetLastError();
                    (void)hipMemRelease(h);
                    ok = place(i, b->handles[i]) == hipSuccess;
                    budget = 0;
                    break;
                }
                const float t = slab_scan_ms(ix, at, slab, d_scratch, ev0, ev1, 3);
                if (t > 0 && t < ms[i]) { // better than what was here: keep it, hold the old one until the end
                    rejects.push_back(b->handles[i]);
                    b->handles[i] = h;
                    ms[i] = t;
                    swapped++;
                    if (t < best)
                        best = t;
                } else { // no better: put the old one back, hold the new one until the end
                    (void)hipMemUnmap(at, slab);
                    ok = place(i, b->handles[i]) == hipSuccess;
                    rejects.push_back(h);
                }
            }
        }
        if (getenv("RLR_ROWS_ALLOC_LOG")) {
            float worst = 0;
            for (float t : ms)
                worst = std::max(worst, t);
            fprintf(stderr, "rlr rows: %zu slabs of %zu MiB, %u fresh slabs tried, %u swapped in; slab scan best %.4f ms worst %.4f ms\n",
                    n_slabs, slab >> 20, tried, swapped, best, worst);
        }
        if (!ok) { // a position could not be re-mapped: give up on this block altogether
            for (auto h : rejects)
                (void)hipMemRelease(h);
            if (ev0) (void)hipEventDestroy(ev0);
            if (ev1) (void)hipEventDestroy(ev1);
            (void)hipFree(d_scratch);
            rows_free(b);
            return hipErrorUnknown;
        }
    }
    for (auto h : rejects)
        (void)hipMemRelease(h);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (d_scratch) (void)hipFree(d_scratch);
    (void)hipGetLastError();
    if (rlr::poison_mode()) {
        (void)hipMemset(b->raw, 0xFF, total);
        (void)hipDeviceSynchronize();
    }
    return hipSuccess;
}

hipError_t rows_alloc(const rlr_index *ix, size_t bytes, RowBlock *b, void **base)
{
    const int device = ix->device;
    static const std::string policy = [] {
        const char *v = getenv("RLR_ROWS_ALLOC");
        return std::string(v ? v : "plain");
    }();
    const char *v = getenv("RLR_ROWS_ALLOC_NOW"); // (re-read per call: the placement experiment creates several indexes in one process)
    const std::string pol = v ? std::string(v) : policy;
    const size_t colon = pol.find(':');
    const std::string kind = pol.substr(0, colon);
    const size_t mib = colon == std::string::npos ? 0 : static_cast<size_t>(strtoull(pol.c_str() + colon + 1, nullptr, 10));
    const size_t A = std::max<size_t>(mib, 2) << 20;
    *b = RowBlock();
    hipError_t e = hipSuccess;
    // ("auto" = "select" from 2 GiB: NOT the default -- see the note above rows_alloc_select)
    if (kind == "select" || (kind == "auto" && bytes >= (2ull << 30))) {
        e = rows_alloc_select(ix, bytes, b, base);
        if (e == hipSuccess)
            return e;
        (void)hipGetLastError();
        *b = RowBlock(); // (no virtual-memory API, or out of memory for the slab rounding: the plain allocation below)
    }
    if (kind == "align") {
        e = rlr::dev_malloc(&b->raw, bytes + A);
        if (e != hipSuccess)
            return e;
        b->raw_bytes = bytes + A;
        *base = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(b->raw) + A - 1) / A * A);
        return hipSuccess;
    }
    if (kind == "round") {
        const size_t r = (bytes + A - 1) / A * A;
        e = rlr::dev_malloc(&b->raw, r);
        b->raw_bytes = r;
        *base = b->raw;
        return e;
    }
    if (kind == "vmm") {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t gran = 0;
        e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        if (e != hipSuccess)
            return e;
        const size_t slab = (A + gran - 1) / gran * gran;
        const size_t total = (bytes + slab - 1) / slab * slab;
        e = hipMemAddressReserve(&b->raw, total, slab, nullptr, 0);
        if (e != hipSuccess)
            return e;
        b->raw_bytes = total;
        b->slab = slab;
        for (size_t off = 0; off < total && e == hipSuccess; off += slab) {
            hipMemGenericAllocationHandle_t h;
            e = hipMemCreate(&h, slab, &prop, 0);
            if (e != hipSuccess)
                break;
            b->handles.push_back(h);
            e = hipMemMap(static_cast<char *>(b->raw) + off, slab, 0, h, 0);
        }
        if (e == hipSuccess) {
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemSetAccess(b->raw, total, &acc, 1);
        }
        if (e != hipSuccess) {
            // (partial mappings: unmap what was mapped, release, free the range)
            for (size_t i = 0; i < b->handles.size(); ++i) {
                (void)hipMemUnmap(static_cast<char *>(b->raw) + i * slab, slab);
                (void)hipMemRelease(b->handles[i]);
            }
            b->handles.clear();
            (void)hipMemAddressFree(b->raw, total);
            *b = RowBlock();
            (void)hipGetLastError();
            return e;
        }
        if (rlr::poison_mode()) {
            (void)hipMemset(b->raw, 0xFF, total);
            (void)hipDeviceSynchronize();
        }
        *base = b->raw;
        return hipSuccess;
    }
    e = rlr::dev_malloc(&b->raw, bytes);
    b->raw_bytes = bytes;
    *base = b->raw;
    return e;
}

int32_t ensure_rows(rlr_index *ix, uint64_t want_rows)
{
    if (want_rows <= ix->cap_rows)
        return RLR_OK;
    if (want_rows > 0xFFFFFFF0ull)
        return fail(RLR_E_INVALID, "an index shard holds at most 2^32-16 rows");
    uint64_t cap = std::max<uint64_t>(want_rows, ix->cap_rows + ix->cap_rows / 2);
    cap = std::max<uint64_t>(cap, 1024);
    void *n = nullptr;
    RowBlock nb;
    hipError_t e = rows_alloc(ix, cap * row_bytes(ix), &nb, &n);
    if (e != hipSuccess && cap > want_rows) {
        cap = want_rows;
        e = rows_alloc(ix, cap * row_bytes(ix), &nb, &n);
    }
    if (e != hipSuccess)
        return fail(RLR_E_OOM, "allocation of %llu rows x %zu B failed: %s",
                    static_cast<unsigned long long>(cap), row_bytes(ix), hipGetErrorString(e));
    if (ix->d_rows && ix->n_rows) {
        hipError_t ce = hipMemcpy(n, ix->d_rows, ix->n_rows * row_bytes(ix), hipMemcpyDeviceToDevice);
        if (ce == hipSuccess)
            ce = hipStreamSynchronize(nullptr); // a device-to-device copy may return before it has run
        if (ce != hipSuccess) {
            rows_free(&nb);
            return fail(RLR_E_HIP, "moving the rows into the larger allocation failed: %s", hipGetErrorString(ce));
        }
    }
    rows_free(&ix->rows_block);
    ix->rows_block = std::move(nb);
    ix->d_rows = n;
    ix->cap_rows = cap;
    if (getenv("RLR_ROWS_ALLOC_LOG"))
        fprintf(stderr, "rlr rows: base %p bytes %zu (raw %p, %zu slabs of %zu)\n", n, static_cast<size_t>(cap * row_bytes(ix)),
                ix->rows_block.raw, ix->rows_block.handles.size(), ix->rows_block.slab);
    return RLR_OK;
}

// (Re)build the 8-bit nomination copy for rows >= first_row; the error / scale maxima only ever grow between
// full rebuilds (a delete keeps the old maxima: conservative).
int32_t sync_q8(rlr_index *ix, uint64_t first_row)
{
    if (!ix->q8_enabled)
        return RLR_OK;
    const uint64_t want = std::max<uint64_t>(ix->cap_rows, ix->n_rows);
    if (ix->q8_cap_rows < want || !ix->d_q8) {
        if (ix->d_q8) (void)hipFree(ix->d_q8);
        if (ix->d_q8_scale) (void)hipFree(ix->d_q8_scale);
        ix->d_q8 = nullptr;
        ix->d_q8_scale = nullptr;
        ix->q8_cap_rows = 0;
        RLR_HIP(rlr::dev_malloc(&ix->d_q8, std::max<uint64_t>(want, 1) * ix->dim));
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&ix->d_q8_scale), std::max<uint64_t>(want, 1) * sizeof(float)));
        ix->q8_cap_rows = want;
        first_row = 0;
    }
    if (!ix->d_q8_stats)
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&ix->d_q8_stats), 4 * sizeof(uint32_t)));
    if (first_row == 0)
        RLR_HIP(hipMemset(ix->d_q8_stats, 0, 4 * sizeof(uint32_t)));
    RLR_HIP(launch_q8_build(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, static_cast<uint32_t>(first_row),
                            static_cast<uint32_t>(ix->n_rows), ix->d_q8, ix->d_q8_scale, ix->d_q8_stats, nullptr));
    uint32_t h[4] = {0, 0, 0, 0};
    RLR_HIP(hipMemcpy(h, ix->d_q8_stats, sizeof(h), hipMemcpyDeviceToHost));
    std::memcpy(&ix->q8_delta, &h[0], 4);
    std::memcpy(&ix->q8_scale_max, &h[1], 4);
    ix->q8_has_inf = h[2] != 0;
    return RLR_OK;
}

// (Re)build the nomination image for every tile that holds a row >= first_row.
int32_t sync_image(rlr_index *ix, uint64_t first_row)
{
    RLR_TRY(sync_q8(ix, first_row));
    if (!ix->image_enabled)
        return RLR_OK;
    const size_t need = image_bytes(ix->dim, std::max<uint64_t>(ix->cap_rows, ix->n_rows));
    if (ix->image_cap < need) {
        if (ix->d_image)
            (void)hipFree(ix->d_image);
        ix->d_image = nullptr;
        ix->image_cap = 0;
        RLR_HIP(rlr::dev_malloc(&ix->d_image, std::max<size_t>(need, 256)));
        ix->image_cap = std::max<size_t>(need, 256);
        first_row = 0; // fresh buffer: every tile has to be written
    }
    const uint32_t t0 = static_cast<uint32_t>(first_row / 256);
    const uint32_t t1 = static_cast<uint32_t>((ix->n_rows + 255) / 256);
    RLR_HIP(launch_build_image(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, static_cast<uint32_t>(ix->n_rows), t0, t1,
                               ix->d_image, nullptr));
    RLR_HIP(hipStreamSynchronize(nullptr));
    return RLR_OK;
}

// Copy host rows in, normalise (optionally) and store them at row `first`.
int32_t ingest(rlr_index *ix, const float *rows, uint64_t n, uint64_t first, int normalize)
{
    if (n == 0)
        return RLR_OK;
    const uint64_t chunk_rows = std::max<uint64_t>(1, (64ull << 20) / (static_cast<uint64_t>(ix->dim) * 4));
    float *d_stage = nullptr, *d_norm = nullptr;
    const uint64_t cr = std::min(chunk_rows, n);
    RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&d_stage), cr * ix->dim * sizeof(float)));
    hipError_t e = rlr::dev_malloc(reinterpret_cast<void **>(&d_norm), cr * sizeof(float));
    if (e != hipSuccess) {
        (void)hipFree(d_stage);
        return fail(RLR_E_OOM, "staging allocation failed");
    }
    int32_t st = RLR_OK;
    std::vector<float> h_norm;
    for (uint64_t r0 = 0; r0 < n && st == RLR_OK; r0 += cr) {
        const uint64_t m = std::min(cr, n - r0);
        e = hipMemcpy(d_stage, rows + r0 * ix->dim, m * ix->dim * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = launch_normalize_store(d_stage, static_cast<uint32_t>(m), ix->dim, normalize,
                                       static_cast<char *>(ix->d_rows) + (first + r0) * row_bytes(ix), ix->pitch16,
                                       ix->dtype, d_norm, nullptr);
        if (e == hipSuccess)
            e = hipStreamSynchronize(nullptr);
        if (e == hipSuccess && !normalize) {
            // rows stored as given: remember the largest norm (NaN rows order last anyway; they do not widen the band)
            h_norm.resize(m);
            e = hipMemcpy(h_norm.data(), d_norm, m * sizeof(float), hipMemcpyDeviceToHost);
            for (uint64_t i = 0; i < m && e == hipSuccess; ++i)
                if (h_norm[i] > ix->max_row_sumsq)
                    ix->max_row_sumsq = h_norm[i];
        }
        if (e != hipSuccess)
            st = fail(RLR_E_HIP, "row ingest failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_stage);
    (void)hipFree(d_norm);
    return st;
}

// ---- waiting for a query without the stream's completion signal ------------------------------------------------
// The last kernel of a pipeline writes the query's candidate count into pinned host memory behind a system-scope fence
// (sort_emit.h), so "the count has left kMetaPending" means "the k results are in host memory".  hipStreamSynchronize
// learns the same thing from the queue's completion signal, which the ROCm runtime waits for in the kernel driver
// (interrupt + wake-up: ~15-20 us between the end of the last kernel and the return, measured as device idle time
// between two searches of a C loop); polling the word takes ~1 us.  RLR_WAIT: "hybrid" (default) polls, but sleeps
// through the first two thirds of a wait that took more than 1.5 ms last time -- a 4.4 ms scan does not burn a core --,
// "spin" always polls from the start, "block" is hipStreamSynchronize.
enum WaitMode { kWaitHybrid = 0, kWaitSpin = 1, kWaitBlock = 2 };

WaitMode wait_mode()
{
    static const WaitMode m = [] {
        const char *v = getenv("RLR_WAIT");
        if (v && v[0] == 's')
            return kWaitSpin;
        if (v && v[0] == 'b')
            return kWaitBlock;
        return kWaitHybrid;
    }();
    return m;
}

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}

// Wait until `complete()` holds -- it reads pinned host words the device writes last (and may check them against the
// data they cover).  The stream is queried every ~1000 polls: an error, or a drained stream with the words still
// incomplete (a kernel that never ran), ends the wait.
template <typename Complete>
int32_t wait_polling(Complete &&complete, hipStream_t s, WaitEma *c, uint32_t key)
{
    using clk = std::chrono::steady_clock;
    const WaitMode mode = wait_mode();
    if (mode == kWaitBlock) {
        RLR_HIP(hipStreamSynchronize(s));
        return RLR_OK;
    }
    const auto t0 = clk::now();
    bool slept = false;
    if (c && c->key != key) {
        c->key = key;
        c->us = 0.0;
    }
    // (only waits of milliseconds: a timed sleep comes back 50-150 us late often enough that at 0.5 ms per query the
    // hybrid form measured 589 us per call where polling from the start took 565 and hipStreamSynchronize 570)
    if (mode == kWaitHybrid && c && c->us > 1500.0) {
        const double sleep_us = 0.7 * c->us - 150.0;
        if (sleep_us > 20.0) {
            timespec ts;
            clock_gettime(CLOCK_MONOTONIC, &ts);
            const long add = static_cast<long>(sleep_us * 1000.0);
            ts.tv_sec += (ts.tv_nsec + add) / 1000000000L;
            ts.tv_nsec = (ts.tv_nsec + add) % 1000000000L;
            (void)clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &ts, nullptr);
            slept = true;
        }
    }
    uint32_t polls = 0;
    for (;;) {
        std::atomic_thread_fence(std::memory_order_acquire);
        if (complete())
            break;
        if ((++polls & 0x3FFu) == 0) {
            const hipError_t e = hipStreamQuery(s);
            if (e == hipSuccess) {
                // drained: everything the kernels stored is on its way; give it a moment, then it is an error
                bool ok = false;
                for (int spin = 0; spin < 100000 && !ok; ++spin) {
                    std::atomic_thread_fence(std::memory_order_acquire);
                    ok = complete();
                    cpu_relax();
                }
                if (ok)
                    break;
                return fail(RLR_E_INTERNAL, "the stream drained but a completion word was never written");
            }
            if (e != hipErrorNotReady)
                return fail(RLR_E_HIP, "stream failed while a search was in flight: %s", hipGetErrorString(e));
        }
        if (polls > 8192 && (polls & 0xFF) == 0)
            std::this_thread::yield(); // a long wait: more waiters than cores must not starve the threads that feed the GPU
        cpu_relax();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (c) {
        const double us = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
        if (slept && polls == 0) // the words were there when the sleep ended: it was too long (a smaller corpus than last time)
            c->us *= 0.5;
        else
            c->us = c->us == 0.0 ? us : 0.75 * c->us + 0.25 * us;
    }
    return RLR_OK;
}

// The result block of a fused search -> MMR call (sort_emit.h: [4 x k_cap values] n status checksum done).
int32_t wait_block(const volatile uint32_t *h_out, uint32_t k_cap, hipStream_t s, WaitEma *c)
{
    return wait_polling(
        [&]() {
            if (h_out[4 * k_cap + 3] != kBlockDone)
                return false;
            const uint32_t n = h_out[4 * k_cap], status = h_out[4 * k_cap + 1];
            if (n > k_cap)
                return false;
            uint32_t chk = 0;
            for (uint32_t b = 0; b < 4; ++b)
                for (uint32_t i = 0; i < n; ++i)
                    chk += result_chk_term(h_out[b * k_cap + i], b * k_cap + i);
            return block_chk_tail(chk, n, status, k_cap) == h_out[4 * k_cap + 2];
        },
        s, c, 0x40000000u);
}

// The completion words of nq single-query pipelines (sort_emit.h: count | checksum << 32, written last).  res != null:
// the k result words of each query are in host memory too and must match the checksum -- the word alone can overtake
// the results on their way through PCIe.
int32_t wait_results(const volatile uint64_t *meta, const volatile uint64_t *res, uint32_t nq, uint32_t k, uint32_t limit,
                     hipStream_t s, WaitEma *c)
{
    uint32_t verified = nq; // queries [verified, nq) are complete (the last one finishes last)
    return wait_polling(
        [&]() {
            while (verified > 0) {
                const uint32_t q = verified - 1;
                const uint64_t m = meta[q];
                if (m == kMetaPending)
                    return false;
                if (res && static_cast<uint32_t>(m) <= limit) { // (an overflowed query emits only its marker: nothing to verify)
                    uint32_t chk = 0;
                    const volatile uint64_t *r = res + static_cast<size_t>(q) * k;
                    for (uint32_t i = 0; i < k; ++i) {
                        const uint64_t w = r[i];
                        if (w)
                            chk += result_chk_term(w, i);
                    }
                    if (chk != static_cast<uint32_t>(m >> 32))
                        return false;
                }
                verified--;
            }
            return true;
        },
        s, c, nq);
}

// ---- query upload -----------------------------------------------------------------------------------------------
// The staged queries (pinned host memory, zero padded to the row pitch) -> the context's device buffer.  A few KB: one
// small kernel on the search's own stream reads them over PCIe itself.  hipMemcpyAsync does the same with the runtime's
// copy kernel, but the scan behind it then starts 4-5 us after that copy has finished (every other boundary of the
// pipeline: < 1 us; scratch/step_timeline.py) -- a stream's copies and kernels are ordered through a signal, kernels
// among themselves by the queue.
} // namespace

namespace rlr {
__global__ __launch_bounds__(256) void stage_query_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, uint32_t n16)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256)
        dst[i] = src[i];
}
} // namespace rlr

namespace {

// Will this index's single-query scans take their query in the kernel arguments?  (Then nothing is uploaded in front of
// them: workgroup 0 of the scan leaves the query in the context's device buffer for the kernels behind it.)
bool scans_take_host_query(const rlr_index *ix, const float *h_q);

// The queries of a call whose pipelines start with enqueue_query_scan: by the scans themselves where they can, else uploaded.
hipError_t upload_queries(Ctx *c, const float *h_q, size_t q_bytes, hipStream_t s);
hipError_t stage_queries_for_scans(const rlr_index *ix, Ctx *c, const float *h_q, size_t q_bytes, hipStream_t s)
{
    c->h_q_kq = scans_take_host_query(ix, h_q) ? h_q : nullptr;
    return c->h_q_kq ? hipSuccess : upload_queries(c, h_q, q_bytes, s);
}

hipError_t upload_queries(Ctx *c, const float *h_q, size_t q_bytes, hipStream_t s)
{
    static const bool by_copy = getenv("RLR_QUERY_MEMCPY") != nullptr; // (A/B)
    if (by_copy || q_bytes > (64u << 10) || (q_bytes & 15) || (reinterpret_cast<uintptr_t>(h_q) & 15))
        return hipMemcpyAsync(c->d_query, h_q, q_bytes, hipMemcpyHostToDevice, s);
    const uint32_t n16 = static_cast<uint32_t>(q_bytes / 16);
    hipLaunchKernelGGL(rlr::stage_query_kernel, dim3(std::min<uint32_t>((n16 + 255) / 256, 8)), dim3(256), 0, s,
                       reinterpret_cast<const float4 *>(h_q), reinterpret_cast<float4 *>(c->d_query), n16);
    return hipGetLastError();
}

// ---- the search pipeline ------------------------------------------------------------
struct SearchPlan {
    uint32_t k;        // per query, already clamped to n_rows
    uint32_t cap;      // candidate capacity
    float two_eps;
    float two_eps_img; // band when the nomination scan ran over the binary16 image
    float scale = 1.0f; // |row|_max * |query|_max when that exceeds 1 (the bands above already carry it)
    bool unordered = false; // the consumer wants the best k as a SET (the hybrid blend re-orders everything anyway)
};

// The guard bands are error bounds for unit-norm operands; every term of them is linear in |row| * |query|.
// Rows stored with normalize_on_device = 0 and queries the caller did not normalise widen the band by that
// product (norms rounded up); unit-norm data -- the reference's invariant, rag_engine.rs:359 / :494 -- gives 1.
// sum of squares in binary64 with eight independent partial sums: the value only feeds an upper bound with its own slack, so
// the order is free -- and one dependent chain per query was 0.2 ms of host time in front of every 256-query batch of 768-d
// queries (1.2 ms in front of config 5's 1024 x 1024-d), more than the whole select-and-finish tail.
static double sumsq_f64(const float *v, uint32_t n)
{
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int j = 0; j < 8; ++j)
            a[j] += static_cast<double>(v[i + j]) * v[i + j];
    for (; i < n; ++i)
        a[0] += static_cast<double>(v[i]) * v[i];
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

float band_scale(const rlr_index *ix, const float *queries, uint32_t nq)
{
    double qmax = 0.0;
    for (uint32_t q = 0; q < nq; ++q) {
        const double s2 = sumsq_f64(queries + static_cast<size_t>(q) * ix->dim, ix->dim);
        if (s2 > qmax) // (a NaN query compares false: its scores are NaN and order last whatever the band)
            qmax = s2;
    }
    const double f = std::sqrt(qmax) * std::sqrt(static_cast<double>(ix->max_row_sumsq)) * 1.000002;
    if (!(f > 1.0001))
        return 1.0f;
    return std::isfinite(f) ? static_cast<float>(f) : 3.0e38f;
}

int32_t ctx_prepare(rlr_index *ix, Ctx *c, uint32_t nq, const SearchPlan &p)
{
    if (c->q_cap < nq || !c->d_query) {
        if (c->d_query) (void)hipFree(c->d_query);
        if (c->d_state) (void)hipFree(c->d_state);
        c->d_query = nullptr;
        c->d_state = nullptr;
        c->q_cap = 0;
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_query), static_cast<size_t>(nq) * ix->q_pitch * sizeof(float)));
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_state), static_cast<size_t>(nq) * sizeof(SelectState)));
        // the fused tail's counters (n_work, done, flags) are zero between two queries; on the context's stream, like the histograms
        RLR_HIP(hipMemsetAsync(c->d_state, 0, static_cast<size_t>(nq) * sizeof(SelectState), c->stream));
        c->q_cap = nq;
    }
    RLR_TRY(grow(&c->d_scores, &c->score_cap, std::max<uint64_t>(ix->n_rows, 4)));
    if (c->cand_cap < p.cap) {
        if (c->d_cand) (void)hipFree(c->d_cand);
        if (c->d_packed) (void)hipFree(c->d_packed);
        c->d_cand = nullptr;
        c->d_packed = nullptr;
        c->cand_cap = 0;
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_cand), static_cast<size_t>(p.cap) * sizeof(uint32_t)));
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_packed), static_cast<size_t>(p.cap) * sizeof(uint64_t)));
        c->cand_cap = p.cap;
    }
    RLR_TRY(grow(&c->d_out, &c->out_cap, static_cast<uint64_t>(nq) * p.k + nq)); // results + per-query counts
    return RLR_OK;
}

} // namespace

namespace rlr {
// Debug assertion (RLR_POISON_ALLOC=1 runs only): every single-query pipeline relies on "the histograms are zero on
// entry" -- established at context creation and re-established by the previous query's re-score kernel, which is only
// stream-order-safe.  A second stream, a reordered launch or a skipped clear would break it silently (a wrong threshold,
// not a crash); this kernel runs in front of every scan in the poisoned test runs and counts non-zero bins into a
// pinned host word that the host checks at the call's synchronisation.
__global__ __launch_bounds__(256) void hist_assert_zero_kernel(const uint32_t *__restrict__ hist, uint32_t n,
                                                               uint32_t *__restrict__ flag)
{
    uint32_t bad = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        bad += hist[i] != 0;
    if (bad)
        atomicAdd(flag, bad);
}

// device-side: sort the re-scored candidates (<= kLdsSortCap) and emit the best k (sort_emit.h).
__global__ __launch_bounds__(1024) void sort_emit_kernel(uint64_t *__restrict__ packed, const SelectState *__restrict__ st,
                                                         uint64_t *__restrict__ out, uint32_t k,
                                                         uint64_t *__restrict__ meta, bool unordered)
{
    __shared__ uint64_t s[4096];
    __shared__ uint32_t s_hist[2048];
    sort_emit_body(packed, st->n_cand, st->cap, out, k, meta, unordered, s, s_hist);
}

// Exchange-step merge: one workgroup per query orders the world x k packed partial
// results the all-gather delivered and emits the global top-k.  Shards are ascending row ranges,
// so (score desc, global row asc) is the same tie rule every shard already applied.
struct MergeBases {
    uint64_t base[16];
};


__global__ __launch_bounds__(1024) void merge_topk_kernel(const uint64_t *__restrict__ gathered, uint32_t world,
                                                         uint32_t n_queries, uint32_t k, MergeBases bases,
                                                         uint64_t *__restrict__ rows_out, float *__restrict__ cos_out,
                                                         uint32_t *__restrict__ n_out, uint64_t *__restrict__ flag_out)
{
    __shared__ uint64_t s[8192];
    __shared__ uint32_t s_valid;
    __shared__ uint32_t s_valid_overflow;
    __shared__ uint32_t s_chk;
    const uint32_t q = blockIdx.x;
    const uint32_t n = world * k;
    uint32_t n_pad = 1;
    while (n_pad < n)
        n_pad <<= 1;
    if (threadIdx.x == 0) {
        s_valid = 0;
        s_valid_overflow = 0;
        s_chk = 0;
    }
    __syncthreads();
    uint32_t valid = 0;
    for (uint32_t i = threadIdx.x; i < n_pad; i += 1024) {
        uint64_t v = 0;
        if (i < n) {
            const uint32_t r = i / k, j = i - r * k;
            const uint64_t p = gathered[(static_cast<size_t>(r) * n_queries + q) * k + j];
            if (p == ~0ull) {
                s_valid_overflow = 1u; // a shard's guard band overflowed: this query has to be redone everywhere
            } else if (p != 0) {
                const uint64_t local = 0xFFFFFFFFull - (p & 0xFFFFFFFFull);
                const uint64_t glob = bases.base[r] + local;
                v = (p & 0xFFFFFFFF00000000ull) | (0xFFFFFFFFull - glob);
                valid++;
            }
        }
        s[i] = v;
    }
    atomicAdd(&s_valid, valid);
    __syncthreads();
    // Every partial list arrives sorted (score desc, row asc; zeros behind) -- that is how sort_emit leaves it -- so an entry's
    // place in the merged order is its index in its own list plus, for every other list, the number of entries there that
    // are greater: world - 1 binary searches of log2(k) steps instead of a sort (a rank sort of the 800 keys of a world-8
    // merge walked ~45 us, a bitonic network of 1024 took ~25 us; this takes ~2).  Keys are unique (the global row is in
    // the low word).  Lists that are NOT sorted (a caller's own data) take the network below.
    __shared__ uint32_t s_unsorted;
    if (threadIdx.x == 0)
        s_unsorted = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i + 1 < n; i += 1024)
        if ((i + 1) % k != 0 && s[i] < s[i + 1])
            s_unsorted = 1u;
    __syncthreads();
    if (!s_unsorted) {
        const uint32_t m_out = min(s_valid, k);
        for (uint32_t i = threadIdx.x; i < n; i += 1024) {
            const uint64_t mine = s[i];
            if (mine == 0)
                continue;
            const uint32_t r = i / k;
            uint32_t rank = i - r * k;
            for (uint32_t o = 0; o < world && rank < k; ++o) {
                if (o == r)
                    continue;
                const uint64_t *lst = s + o * k;
                uint32_t lo = 0, hi = k; // first position whose entry is not greater than `mine`
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (lst[mid] > mine)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                rank += lo;
            }
            if (rank < m_out) {
                const uint64_t row = 0xFFFFFFFFull - (mine & 0xFFFFFFFFull);
                const float cs = key_score(static_cast<uint32_t>(mine >> 32));
                rows_out[static_cast<size_t>(q) * k + rank] = row;
                cos_out[static_cast<size_t>(q) * k + rank] = cs;
                atomicAdd(&s_chk, result_chk_term(row, rank) + result_chk_term(__builtin_bit_cast(uint32_t, cs), rank + k));
            }
        }
        for (uint32_t i = m_out + threadIdx.x; i < k; i += 1024) { // (the padding behind fewer than k results)
            const float cs = key_score(0u);
            rows_out[static_cast<size_t>(q) * k + i] = ~0ull;
            cos_out[static_cast<size_t>(q) * k + i] = cs;
            atomicAdd(&s_chk, result_chk_term(~0ull, i) + result_chk_term(__builtin_bit_cast(uint32_t, cs), i + k));
        }
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t cnt = s_valid_overflow ? 0xFFFFFFFFu : m_out;
            n_out[q] = cnt;
            if (flag_out)
                flag_out[q] = (static_cast<uint64_t>(s_chk) << 32) | cnt;
        }
        return;
    } else
    for (uint32_t kk = 2; kk <= n_pad; kk <<= 1) {
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n_pad; i += 1024) {
                const uint32_t ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = s[i], b = s[ixj];
                    const bool desc = (i & kk) == 0;
                    if (desc ? (a < b) : (a > b)) {
                        s[i] = b;
                        s[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    const uint32_t m = min(s_valid, k);
    for (uint32_t i = threadIdx.x; i < k; i += 1024) {
        const uint64_t v = i < m ? s[i] : 0ull;
        const uint64_t row = i < m ? 0xFFFFFFFFull - (v & 0xFFFFFFFFull) : ~0ull;
        const float cs = key_score(static_cast<uint32_t>(v >> 32));
        rows_out[static_cast<size_t>(q) * k + i] = row;
        cos_out[static_cast<size_t>(q) * k + i] = cs;
        atomicAdd(&s_chk, result_chk_term(row, i) + result_chk_term(__builtin_bit_cast(uint32_t, cs), i + k));
    }
    // count | checksum << 32 last, behind a system-scope fence: a host that pre-set the word to kMetaPending polls it and
    // verifies the checksum over the rows and scores it finds in its memory (sort_emit.h: the word can overtake them)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t cnt = s_valid_overflow ? 0xFFFFFFFFu : m;
        n_out[q] = cnt;
        if (flag_out)
            flag_out[q] = (static_cast<uint64_t>(s_chk) << 32) | cnt;
    }
}

// The large-candidate path's finish (a guard band that outgrew the 4096-entry LDS sort: dense score distributions under
// the 8-bit nomination, k in the thousands): the k best of n exactly re-scored keys in global memory by one workgroup --
// radix select of the k-th key over the L2-resident list (3-6 passes, histograms in LDS), the winners gathered into LDS,
// sorted, emitted.  One launch where a global bitonic network took log^2(n) of them.
__global__ __launch_bounds__(1024) void topk_global_kernel(const uint64_t *__restrict__ keys, const SelectState *__restrict__ st,
                                                           uint32_t cap, uint64_t *__restrict__ out, uint32_t k)
{
    __shared__ uint64_t s[4096];
    __shared__ uint32_t s_hist[2048];
    __shared__ uint32_t s_pick[3];
    __shared__ uint32_t s_n;
    const uint32_t n = min(st->n_cand, cap);
    if (threadIdx.x == 0)
        s_n = 0;
    __syncthreads();
    uint64_t kth = 0;
    if (n > k)
        kth = lds_kth_key64(keys, n, k, s_hist, s_pick, 1024); // (the helper only needs a pointer; keys are unique)
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const uint64_t v = keys[i];
        if (v >= kth && v != 0) {
            const uint32_t at = atomicAdd(&s_n, 1u);
            if (at < 4096)
                s[at] = v;
        }
    }
    __syncthreads();
    const uint32_t m = min(s_n, 4096u);
    uint32_t n_pad = 1;
    while (n_pad < m)
        n_pad <<= 1;
    for (uint32_t i = m + threadIdx.x; i < n_pad; i += 1024)
        s[i] = 0;
    __syncthreads();
    for (uint32_t kk = 2; kk <= n_pad; kk <<= 1)
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n_pad; i += 1024) {
                const uint32_t ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = s[i], b = s[ixj];
                    const bool desc = (i & kk) == 0;
                    if (desc ? (a < b) : (a > b)) {
                        s[i] = b;
                        s[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < k; i += 1024)
        out[i] = i < m ? s[i] : 0ull;
}

__global__ void emit_kernel(const uint64_t *__restrict__ packed, uint32_t n, uint64_t *__restrict__ out, uint32_t k)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k)
        out[i] = i < n ? packed[i] : 0ull;
}

// ---- search -> MMR without a host round trip (rlr_search_diverse): pool_prepare.h ---------------------------
template <bool FROM_CANDIDATES>
__global__ __launch_bounds__(1024) void pool_prepare_kernel(const uint64_t *__restrict__ packed, const SelectState *__restrict__ st,
                                                            PoolArgs pa)
{
    __shared__ __attribute__((aligned(16))) char s_pool[kPoolLdsBytes];
    pool_prepare_body<FROM_CANDIDATES, false>(packed, FROM_CANDIDATES ? st->n_cand : 0u, FROM_CANDIDATES ? st->cap : 0u, pa, s_pool);
}


// ---- hybrid search without host round trips (rlr_search_hybrid) -------------------------------------------
// The device twin of search_impl's candidate list when lexical (BM25) candidates exist (csrc/engine.cpp;
// rag_engine.rs:505-561): candidates = the `fetch` best rows by cosine UNITED with the lexical rows,
// combined = w_e * cos + w_l * (lex / max_lex) (two rounded products, one add; IEEE division), ordered
// (combined desc, NaN last, row asc), cut to `need`.  Slots [0, fetch) hold the fetched rows, [fetch, fetch + n_lex)
// the lexical rows that were not fetched; one bitonic sort in LDS orders them.  The same boundary rule as the host:
// every unfetched row is non-lexical and scores at most combine(cos of the last fetched row, 0); if the need-th
// candidate does not beat that, info[1] = 2 and the host takes the widening path.
constexpr uint32_t kHybridSlots = 4096, kHybridLexMax = 2048, kHybridSelMax = 2048, kHybridHash = 4096;
constexpr uint32_t kHybridFetchMargin = 32; // rows by cosine fetched beyond `need` (rlr_index::hybrid_fetch_full): room for ties

// What the blend needs to know about the lexical pairs, in device memory: written by the host copy (pairs handed in by
// the caller) or by lex_unpack_kernel (pairs left on the device by a BM25 scoring call).
struct HybridLexHeader {
    uint32_t n_lex;
    float max_lex; // max(lexical scores, f32::EPSILON) (rag_engine.rs:515-519)
};

// A scoring call's result (pack_result(score, row) keys, in any order) -> rows, scores, header.  Rows outside the index
// (a lexical index that ran ahead of the embedding matrix) keep their place but are marked: they still count for
// max_lexical, as in the reference, and never become candidates.  One workgroup (<= kHybridLexMax pairs).
__global__ __launch_bounds__(1024) void lex_unpack_kernel(const uint64_t *__restrict__ packed, const uint32_t *__restrict__ count,
                                                          uint32_t limit, uint32_t n_rows, uint32_t *__restrict__ lrow,
                                                          float *__restrict__ lscore, HybridLexHeader *__restrict__ hdr)
{
    __shared__ uint32_t s_max;
    if (threadIdx.x == 0)
        s_max = 0;
    __syncthreads();
    if (*count == kLexicalRetry) { // the BM25 selection handed the query back: so does the blend (info[1] = 3)
        if (threadIdx.x == 0) {
            hdr->n_lex = kLexicalRetry;
            hdr->max_lex = 1.0f;
        }
        return;
    }
    const uint32_t n = min(*count, limit);
    uint32_t mx = 0; // largest ordered score key: fold(0.0, f32::max) over the scores (:515-519)
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        float sc;
        uint32_t row;
        unpack_result(packed[i], &sc, &row);
        lrow[i] = row < n_rows ? row : 0xFFFFFFFFu;
        lscore[i] = sc;
        mx = max(mx, static_cast<uint32_t>(packed[i] >> 32));
    }
    atomicMax(&s_max, mx);
    __syncthreads();
    if (threadIdx.x == 0) {
        hdr->n_lex = n;
        const float m = s_max ? fmaxf(0.0f, key_score(s_max)) : 0.0f; // NaN scores (key 0) are ignored like f32::max does
        hdr->max_lex = m >= 1.1920929e-07f ? m : 1.1920929e-07f;
    }
}

// [row u32 | cos f32 | combined f32 | lexical f32] x k_cap, then n, status, checksum, done (the layout the greedy kernel's
// emit tail writes) into pinned host memory, by all threads of one workgroup
__device__ inline void hybrid_emit_body(const uint32_t *list, const float *comb, const float *cosv, const float *lexv, uint32_t n,
                                        uint32_t status, uint32_t k_cap, uint32_t *__restrict__ h_out)
{
    __shared__ uint32_t s_chk;
    if (threadIdx.x == 0)
        s_chk = 0;
    __syncthreads();
    uint32_t chk = 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t w0 = list[i], w1 = __builtin_bit_cast(uint32_t, cosv[i]), w2 = __builtin_bit_cast(uint32_t, comb[i]);
        const uint32_t w3 = __builtin_bit_cast(uint32_t, lexv[i]);
        h_out[i] = w0;
        h_out[k_cap + i] = w1;
        h_out[2 * k_cap + i] = w2;
        h_out[3 * k_cap + i] = w3;
        chk += result_chk_term(w0, i) + result_chk_term(w1, k_cap + i) + result_chk_term(w2, 2 * k_cap + i) +
               result_chk_term(w3, 3 * k_cap + i);
    }
    if (chk)
        atomicAdd(&s_chk, chk);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { // (the greedy kernel's emit tail writes the same four words)
        h_out[4 * k_cap] = n;
        h_out[4 * k_cap + 1] = status;
        h_out[4 * k_cap + 2] = block_chk_tail(s_chk, n, status, k_cap);
        __threadfence_system();
        h_out[4 * k_cap + 3] = kBlockDone;
    }
}

__global__ __launch_bounds__(1024) void hybrid_pool_kernel(const uint64_t *__restrict__ packed, uint32_t fetch, uint32_t need,
                                                           uint32_t n_rows, float w_e, float w_l,
                                                           const uint32_t *__restrict__ lrow, const float *__restrict__ lscore,
                                                           const float *__restrict__ lcos,
                                                           const HybridLexHeader *__restrict__ hdr,
                                                           float *__restrict__ cand, // 3 x kHybridSlots: combined | cos | lex
                                                           uint32_t *__restrict__ list, float *__restrict__ comb,
                                                           float *__restrict__ cosv, float *__restrict__ lexv,
                                                           uint32_t *__restrict__ info,
                                                           uint32_t k_cap, uint32_t *__restrict__ h_out) // h_out != null: a search
                                                           // without diversification -- the pool IS the result, emitted here
{
    __shared__ uint64_t s_key[kHybridSlots];   // slot -> (ordered combined score, ~row); 0 = empty
    __shared__ uint64_t s_sel[kHybridSelMax];  // the keys that can be among the first `need`
    __shared__ uint32_t s_slot[kHybridSelMax]; // ... and their slots
    __shared__ uint32_t s_hrow[kHybridHash];   // open-addressing map lexical row + 1 -> its index in the pair list
    __shared__ uint32_t s_hidx[kHybridHash];
    __shared__ uint32_t s_flag[kHybridLexMax]; // lexical pair j was reached by the fetch
    __shared__ uint32_t s_hist[2048];
    __shared__ uint32_t s_pick[3];
    __shared__ uint32_t s_got, s_cand, s_nsel, s_emin;
    __shared__ float s_cneed;
    const uint32_t t = threadIdx.x;
    float *cc = cand, *ce = cand + kHybridSlots, *cl = cand + 2 * kHybridSlots;
    if (hdr->n_lex == kLexicalRetry) { // (uniform) no usable lexical pairs: the host path repeats the BM25 query exactly
        if (threadIdx.x == 0) {
            info[0] = 0;
            info[1] = 3;
        }
        if (h_out)
            hybrid_emit_body(list, comb, cosv, lexv, 0u, 3u, k_cap, h_out);
        return;
    }
    const uint32_t n_lex = min(hdr->n_lex, kHybridLexMax);
    const float max_lex = hdr->max_lex;
    if (t == 0) {
        s_got = 0;
        s_cand = 0;
        s_nsel = 0;
        s_cneed = 0.0f;
        s_emin = 0xFFFFFFFFu;
    }
    for (uint32_t i = t; i < kHybridHash; i += 1024)
        s_hrow[i] = 0;
    for (uint32_t j = t; j < n_lex; j += 1024)
        s_flag[j] = 0;
    __syncthreads();
    for (uint32_t j = t; j < n_lex; j += 1024) {
        const uint32_t row = lrow[j];
        if (row == 0xFFFFFFFFu)
            continue;
        uint32_t h = (row * 2654435761u) >> 20; // 12 bits
        for (;;) {
            const uint32_t old = atomicCAS(&s_hrow[h], 0u, row + 1);
            if (old == 0u) {
                s_hidx[h] = j;
                break;
            }
            h = (h + 1) & (kHybridHash - 1); // rows are unique: never the same key twice
        }
    }
    __syncthreads();
    const bool overflow = packed[0] == ~0ull;
    const uint32_t total = fetch + n_lex;
    for (uint32_t i = t; i < fetch; i += 1024) {
        const uint64_t p = overflow ? 0ull : packed[i];
        uint64_t key = 0;
        if (p != 0) { // valid entries are a prefix (in no particular order), padding zeros behind
            const uint32_t row = 0xFFFFFFFFu - static_cast<uint32_t>(p & 0xFFFFFFFFull);
            const float e = key_score(static_cast<uint32_t>(p >> 32));
            atomicMin(&s_emin, static_cast<uint32_t>(p >> 32)); // the smallest fetched cosine (NaN orders lowest)
            float l = 0.0f;
            uint32_t h = (row * 2654435761u) >> 20;
            for (;;) {
                const uint32_t hr = s_hrow[h];
                if (hr == 0u)
                    break;
                if (hr == row + 1) {
                    const uint32_t j = s_hidx[h];
                    l = lscore[j] / max_lex;
                    s_flag[j] = 1;
                    break;
                }
                h = (h + 1) & (kHybridHash - 1);
            }
            const float t0 = w_e * e;
            const float t1 = w_l * l;
            const float c = t0 + t1;
            cc[i] = c;
            ce[i] = e;
            cl[i] = l;
            key = (static_cast<uint64_t>(score_key(c == 0.0f ? 0.0f : c)) << 32) | (p & 0xFFFFFFFFull); // -0 ties with +0
            atomicAdd(&s_got, 1u);
        }
        s_key[i] = key;
    }
    __syncthreads();
    for (uint32_t j = t; j < n_lex; j += 1024) {
        uint64_t key = 0;
        const uint32_t slot = fetch + j;
        const uint32_t row = lrow[j];
        if (row != 0xFFFFFFFFu && !s_flag[j]) { // a lexical row the fetch did not reach
            const float e = lcos[j];
            const float l = lscore[j] / max_lex;
            const float t0 = w_e * e;
            const float t1 = w_l * l;
            const float c = t0 + t1;
            cc[slot] = c;
            ce[slot] = e;
            cl[slot] = l;
            key = (static_cast<uint64_t>(score_key(c == 0.0f ? 0.0f : c)) << 32) | (0xFFFFFFFFu - row);
            atomicAdd(&s_cand, 1u);
        }
        s_key[slot] = key;
    }
    __syncthreads();
    const uint32_t got = s_got, n_cand = got + s_cand, n_pool = min(n_cand, need);
    // a lower edge that between need and need + 32 of the full keys reach (the passes of the radix select stop as soon as a
    // bin decides that much), then those are ranked.  [It was the need-th largest 32-bit SCORE in three fixed passes and
    // everything at or above it: a score shared by thousands of candidates flooded the rank sort and sent the query to the
    // host path -- the row half of the key splits such a class here.]
    uint64_t key_lo = 0;
    if (n_cand > need)
        key_lo = lds_kth_key64(s_key, total, need, s_hist, s_pick, 1024, /*slack=*/32);
    for (uint32_t i = t; i < total; i += 1024) {
        const uint64_t v = s_key[i];
        if (v != 0 && v >= key_lo) {
            const uint32_t at = atomicAdd(&s_nsel, 1u);
            if (at < kHybridSelMax) {
                s_sel[at] = v;
                s_slot[at] = i;
            }
        }
    }
    __syncthreads();
    const uint32_t n_sel = s_nsel;
    const bool flood = n_sel > kHybridSelMax; // (cannot happen with the select above; kept as the guard of s_sel's capacity)
    if (!flood)
        for (uint32_t i = t; i < n_sel; i += 1024) {
            const uint64_t mine = s_sel[i];
            const uint32_t rank = lds_rank_desc(s_sel, n_sel, mine); // keys are unique (the row is part of the key)
            if (rank < n_pool) {
                const uint32_t slot = s_slot[i];
                list[rank] = 0xFFFFFFFFu - static_cast<uint32_t>(mine & 0xFFFFFFFFull);
                comb[rank] = cc[slot];
                cosv[rank] = ce[slot];
                lexv[rank] = cl[slot];
                if (rank == need - 1)
                    s_cneed = cc[slot];
            }
        }
    for (uint32_t r = n_pool + t; r < need; r += 1024) { // unused slots: a valid row, never read by the greedy kernel
        list[r] = 0;
        comb[r] = 0.0f;
        cosv[r] = 0.0f;
        lexv[r] = 0.0f;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t status = overflow ? 1u : flood ? 2u : 0u;
        if (!status && got < n_rows && got > 0) {
            const float t0 = w_e * key_score(s_emin); // the smallest fetched cosine bounds every unfetched (non-lexical) row
            const float t1 = w_l * 0.0f;
            const float c_tail = t0 + t1;
            const bool ok = n_cand >= need && (c_tail != c_tail || s_cneed > c_tail);
            if (!ok)
                status = 2u;
        }
        info[0] = status ? 0u : n_pool;
        info[1] = status;
        s_pick[0] = status;
    }
    if (h_out) { // (list .. lexv were stored by this workgroup's own threads in front of the barrier above)
        __syncthreads();
        const uint32_t status = s_pick[0];
        hybrid_emit_body(list, comb, cosv, lexv, status ? 0u : min(n_pool, k_cap), status, k_cap, h_out);
    }
}

// the first info[0] candidates in their sorted order (no diversification) -> pinned host memory: hybrid_emit_body, below
// hybrid_pool_kernel's helpers; this launch only exists for RLR_HYBRID_EMIT=split (the blend kernel emits by itself)
__global__ __launch_bounds__(256) void hybrid_emit_kernel(const uint32_t *__restrict__ list, const float *__restrict__ comb,
                                                          const float *__restrict__ cosv, const float *__restrict__ lexv,
                                                          const uint32_t *__restrict__ info, uint32_t k_cap,
                                                          uint32_t *__restrict__ h_out)
{
    hybrid_emit_body(list, comb, cosv, lexv, info[1] ? 0u : min(info[0], k_cap), info[1], k_cap, h_out);
}
} // namespace rlr

namespace {

// f32 rows with an up-to-date image and the opt-in set: the nomination scan reads the binary16 image
bool scan_over_image(const rlr_index *ix)
{
    return ix->image_scan && ix->image_enabled && ix->d_image && ix->dtype == RLR_F32;
}

// ||q||_2 per staged query, rounded up (only the 8-bit nomination band needs it)
void stage_query_norms(const rlr_index *ix, Ctx *c, const float *queries, uint32_t nq)
{
    if (!ix->q8_enabled)
        return;
    c->q_norm.resize(nq);
    for (uint32_t q = 0; q < nq; ++q) {
        const double s2 = sumsq_f64(queries + static_cast<size_t>(q) * ix->dim, ix->dim);
        c->q_norm[q] = std::isfinite(s2) ? static_cast<float>(std::sqrt(s2) * 1.000001) : 1.0f;
    }
}

// f32 rows with an up-to-date 8-bit copy: the nomination scan reads one byte per element
bool scan_over_q8(const rlr_index *ix)
{
    return ix->q8_enabled && ix->d_q8 && !ix->q8_has_inf;
}

bool scans_take_host_query(const rlr_index *ix, const float *h_q)
{
    if (scan_over_q8(ix) || scan_over_image(ix) || ix->n_rows == 0)
        return false;
    ScanArgs sa;
    sa.rows = ix->d_rows;
    sa.query = nullptr;
    sa.scores = nullptr;
    sa.hist = nullptr;
    sa.n_rows = static_cast<uint32_t>(ix->n_rows);
    sa.dim = ix->dim;
    sa.pitch16 = ix->pitch16;
    sa.dtype = ix->dtype;
    sa.n_cu = ix->n_cu;
    sa.variant = ix->scan_variant;
    sa.query_host = h_q;
    return launch_scan_takes_host_query(sa);
}

// band for 8-bit-nominated scores of a query of norm q_norm (Cauchy-Schwarz on the stored row error norms)
float q8_two_eps(const rlr_index *ix, float q_norm, float guard_eps)
{
    const float dflt = rlr_default_guard_eps(ix->dim);
    const float scale = guard_eps > dflt ? guard_eps / dflt : 1.0f;
    const float qn = q_norm * 1.0001f + 1e-30f;
    return 2.0f * (ix->q8_delta * qn + q8_arith_eps(ix->dim, ix->q8_scale_max, qn) + dflt) * scale;
}

// band for image-nominated scores: the nomination bound, scaled like the caller scaled guard_eps
float image_two_eps(const rlr_index *ix, float guard_eps)
{
    const float dflt = rlr_default_guard_eps(ix->dim);
    const float scale = guard_eps > dflt ? guard_eps / dflt : 1.0f;
    return 2.0f * nomination_eps(ix->dim, ix->dtype) * scale;
}

// after a synchronisation: did the zero-histogram assertion of a poisoned run fire?
int32_t check_hist_assert(Ctx *c)
{
    if (c->h_assert && *c->h_assert) {
        const uint32_t n = *c->h_assert;
        *c->h_assert = 0;
        return fail(RLR_E_INTERNAL, "zero-histogram invariant violated: %u non-zero bins in front of a scan", n);
    }
    return RLR_OK;
}

// Enqueue the whole pipeline for query `qi` on the context's stream:
//   scan (+digit-1 histogram) -> the tail in two launches (tail.hip), or in its split form: digit-2 histogram (bin search
//   folded in) -> collect (bin search folded in) -> LDS-staged reference-order re-score -> sort + emit; either way the
//   candidate count (| checksum) goes into *d_meta_q last.
// The context's histograms are zero on entry: cleared at creation and by every pipeline (tail stage 2 / the re-score kernel).
// (in two halves, so that a caller with work for ANOTHER stream -- the BM25 kernels of a text search -- can launch it right
// behind the scan instead of behind every launch of the pipeline: enqueue_query_scan, then enqueue_query_rest)
hipError_t enqueue_query_scan(rlr_index *ix, Ctx *c, uint32_t qi, bool timed)
{
    hipStream_t s = c->stream;
    hipError_t e;
    const uint32_t n = static_cast<uint32_t>(ix->n_rows);
    uint32_t *hist1 = c->d_hist;
    const float *dq = c->d_query + static_cast<size_t>(qi) * ix->q_pitch;

    if (c->h_assert) {
        hipLaunchKernelGGL(rlr::hist_assert_zero_kernel, dim3(4), dim3(256), 0, s, c->d_hist, 2u * kHistBins, c->h_assert);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (timed && (e = hipEventRecord(c->ev[0], s)) != hipSuccess) return e;
    ScanArgs sa;
    sa.rows = ix->d_rows;
    sa.query = dq;
    sa.scores = c->d_scores;
    sa.hist = hist1;
    sa.n_rows = n;
    sa.dim = ix->dim;
    sa.pitch16 = ix->pitch16;
    sa.dtype = ix->dtype;
    sa.n_cu = ix->n_cu;
    sa.variant = ix->scan_variant;
    sa.query_host = c->h_q_kq ? c->h_q_kq + static_cast<size_t>(qi) * ix->q_pitch : nullptr;
    static const bool variant_dyn = getenv("RLR_SCAN_VARIANT_DYN") != nullptr; // (experiments: the variant re-read per launch)
    if (variant_dyn)
        if (const char *v = getenv("RLR_SCAN_VARIANT"))
            sa.variant = static_cast<int>(strtol(v, nullptr, 0));
    const bool q8 = scan_over_q8(ix);
    const bool img = !q8 && scan_over_image(ix);
    if (q8)
        e = launch_q8_scan(ix->d_q8, ix->d_q8_scale, n, ix->dim, dq, c->d_scores, hist1, ix->n_cu, s);
    else if (img)
        e = launch_scan_image(ix->d_image, n, ix->dim, dq, c->d_scores, hist1, ix->n_cu, s);
    else
        e = launch_scan(sa, s);
    if (e != hipSuccess) return e;
    if (timed && (e = hipEventRecord(c->ev[1], s)) != hipSuccess) return e;
    return hipSuccess;
}

// pool != nullptr (with emit == false): a diversified search -- when the fused tail runs, its finish builds the MMR pool in
// the same launch and *pool_done = true; otherwise the caller launches pool_prepare_kernel itself.
hipError_t enqueue_query_rest(rlr_index *ix, Ctx *c, uint32_t qi, const SearchPlan &p, uint64_t *d_out_q, uint64_t *d_meta_q,
                              bool timed, bool emit = true, const PoolArgs *pool = nullptr, bool *pool_done = nullptr)
{
    if (pool_done)
        *pool_done = false;
    hipStream_t s = c->stream;
    hipError_t e;
    const uint32_t n = static_cast<uint32_t>(ix->n_rows);
    uint32_t *hist1 = c->d_hist, *hist2 = c->d_hist + kHistBins;
    SelectState *st = c->d_state + qi;
    const float *dq = c->d_query + static_cast<size_t>(qi) * ix->q_pitch;
    const bool q8 = scan_over_q8(ix);
    const bool img = !q8 && scan_over_image(ix);
    const float band = q8 ? q8_two_eps(ix, qi < c->q_norm.size() ? c->q_norm[qi] : 1.0f, p.two_eps * 0.5f)
                          : (img ? p.two_eps_img : p.two_eps);
    // Up to a few million rows the k-th score's digit-1 bin holds a few hundred scores and the tail's one-pass (DIRECT)
    // mode applies: 15 + 8 us at 1.25 M rows where the four launches took 30 + three launch boundaries.  At 10 M rows the
    // bin holds thousands, both forms make two passes over the 40 MB of scores, and the four specialised kernels win
    // (31 us against 15 + 27: their histogram / collect passes run at a third of the registers and LDS).
    const bool fused = ix->fused_tail < 0 ? ix->n_rows <= 4000000ull : ix->fused_tail != 0;
    if (fused && p.cap <= kLdsSortCap && tail_fits(ix->pitch16, ix->dim, ix->dtype)) {
        // two launches (tail.hip): bin search + collect + re-score (or the digit-2 histogram of a crowded bin), then sort + emit
        TailArgs ta;
        ta.scores = c->d_scores;
        ta.n = n;
        ta.hist = c->d_hist;
        ta.st = st;
        ta.k = p.k;
        ta.cap = p.cap;
        ta.two_eps = band;
        ta.rows = ix->d_rows;
        ta.pitch16 = ix->pitch16;
        ta.dim = ix->dim;
        ta.dtype = ix->dtype;
        ta.query = dq;
        ta.packed = c->d_packed;
        ta.out = emit ? d_out_q : nullptr;
        ta.meta = d_meta_q;
        ta.unordered = p.unordered;
        ta.pool = emit ? nullptr : pool;
        ta.direct_max = ta.pool ? std::min<uint32_t>(ix->tail_direct_max, 1024u) : ix->tail_direct_max;
        ta.n_cu = ix->n_cu;
        if (pool_done)
            *pool_done = ta.pool != nullptr;
        if ((e = launch_tail_stage1(ta, s)) != hipSuccess) return e;
        if (timed && (e = hipEventRecord(c->ev[2], s)) != hipSuccess) return e;
        if ((e = launch_tail_stage2(ta, s)) != hipSuccess) return e;
        if (timed && (e = hipEventRecord(c->ev[3], s)) != hipSuccess) return e;
        return hipSuccess;
    }
    if ((e = launch_hist2_find1(c->d_scores, n, hist1, hist2, st, p.k, p.cap, ix->n_cu, s)) != hipSuccess) return e;
    if ((e = launch_collect_find2(c->d_scores, n, hist2, st, band, c->d_cand, ix->n_cu, s)) != hipSuccess)
        return e;
    if (timed && (e = hipEventRecord(c->ev[2], s)) != hipSuccess) return e;
    const uint32_t n_max = std::min<uint32_t>(p.cap, kLdsSortCap);
    if (!launch_rescore_staged(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, dq, c->d_cand, st, c->d_packed, n_max,
                               c->d_hist, s, &e)) {
        // rows too large for the staged layout: one lane per candidate, then clear the histograms
        if ((e = launch_rescore(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, dq, c->d_cand, st, c->d_packed, n_max, s)) !=
            hipSuccess)
            return e;
        e = hipMemsetAsync(c->d_hist, 0, 2 * kHistBins * sizeof(uint32_t), s);
    }
    if (e != hipSuccess) return e;
    if (emit) { // (a caller that orders the re-scored candidates itself -- pool_prepare_kernel<true> -- skips this launch)
        hipLaunchKernelGGL(rlr::sort_emit_kernel, dim3(1), dim3(1024), 0, s, c->d_packed, st, d_out_q, p.k, d_meta_q, p.unordered);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (timed && (e = hipEventRecord(c->ev[3], s)) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t enqueue_query(rlr_index *ix, Ctx *c, uint32_t qi, const SearchPlan &p, uint64_t *d_out_q, uint64_t *d_meta_q,
                         bool timed, bool emit = true, const PoolArgs *pool = nullptr, bool *pool_done = nullptr)
{
    const hipError_t e = enqueue_query_scan(ix, c, qi, timed);
    return e != hipSuccess ? e : enqueue_query_rest(ix, c, qi, p, d_out_q, d_meta_q, timed, emit, pool, pool_done);
}

// Large-candidate path for one query whose band overflowed the LDS sort (massive ties /
// duplicated chunks, or k > kLdsSortCap).  Re-uses the scores still resident from the scan
// only when the query was the last one scanned; otherwise re-scans.
int32_t big_query(rlr_index *ix, Ctx *c, uint32_t qi, const SearchPlan &p, uint32_t n_cand, bool rescan,
                  uint64_t *d_out_q)
{
    hipStream_t s = c->stream;
    const uint32_t n = static_cast<uint32_t>(ix->n_rows);
    const uint32_t cap = next_pow2(std::max<uint32_t>(n_cand, p.k));
    SearchPlan big = p;
    big.cap = cap;
    if (c->cand_cap < cap) {
        if (c->d_cand) (void)hipFree(c->d_cand);
        if (c->d_packed) (void)hipFree(c->d_packed);
        c->d_cand = nullptr;
        c->d_packed = nullptr;
        c->cand_cap = 0;
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_cand), static_cast<size_t>(cap) * sizeof(uint32_t)));
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_packed), static_cast<size_t>(cap) * sizeof(uint64_t)));
        c->cand_cap = cap;
    }
    SelectState *st = c->d_state + qi;
    const float *dq = c->d_query + static_cast<size_t>(qi) * ix->q_pitch;
    if (rescan) {
        ScanArgs sa;
        sa.rows = ix->d_rows;
        sa.query = dq;
        sa.scores = c->d_scores;
        sa.hist = nullptr;
        sa.n_rows = n;
        sa.dim = ix->dim;
        sa.pitch16 = ix->pitch16;
        sa.dtype = ix->dtype;
        sa.n_cu = ix->n_cu;
        sa.variant = ix->scan_variant;
        RLR_HIP(launch_scan(sa, s));
    }
    // key_lo is still valid in the state; reset the counter and the capacity
    SelectState h;
    RLR_HIP(hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, s));
    RLR_HIP(hipStreamSynchronize(s));
    h.n_cand = 0;
    h.cap = cap;
    RLR_HIP(hipMemcpyAsync(st, &h, sizeof(h), hipMemcpyHostToDevice, s));
    RLR_HIP(launch_collect(c->d_scores, n, st, c->d_cand, ix->n_cu, s));
    static const bool old_finish = getenv("RLR_BIG_QUERY_SORT") != nullptr; // A/B and test switch: the global bitonic sort
    hipError_t e = hipSuccess;
    const bool second_level = !old_finish && p.k <= kLdsSortCap;
    // (the staged kernel writes the n_cand keys only; the global sort below needs the zero padding up to `cap` that the
    // one-lane kernel writes)
    const bool staged = second_level && launch_rescore_staged(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, dq, c->d_cand, st,
                                                              c->d_packed, cap, nullptr, s, &e);
    RLR_HIP(e);
    if (!staged) // rows too large for the staged layout, k beyond the one-workgroup finish, or the switch
        RLR_HIP(launch_rescore(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, dq, c->d_cand, st, c->d_packed, cap, s));
    if (second_level) {
        // second level: the exact keys are selected and sorted by one workgroup
        hipLaunchKernelGGL(rlr::topk_global_kernel, dim3(1), dim3(1024), 0, s, c->d_packed, st, cap, d_out_q, p.k);
    } else {
        RLR_HIP(launch_sort_desc(c->d_packed, cap, s));
        hipLaunchKernelGGL(rlr::emit_kernel, dim3((p.k + 255) / 256), dim3(256), 0, s, c->d_packed, n_cand, d_out_q, p.k);
    }
    RLR_HIP(hipGetLastError());
    RLR_HIP(hipStreamSynchronize(s));
    return RLR_OK;
}

// ---- batched (matrix-core) pipeline ---------------------------------------------------
constexpr uint32_t kBatchMaxQueries = 1024; // queries per batched pipeline run (bounds the workspace)

bool env_is_one(const char *name)
{
    const char *v = getenv(name);
    return v && v[0] == '1';
}

bool batch_eligible(const rlr_index *ix, uint32_t nq, uint32_t k)
{
    if (nq < 2 || ix->dim % 128 != 0 || ix->n_rows < 4096 || k * 8 > batch_finish_capacity())
        return false;
    if (ix->batch_min > 0)
        return nq >= ix->batch_min;
    // Cost model from the measured rates: a single-query pipeline streams the
    // rows at ~6.5 TB/s plus ~60 us of fixed cost; a batch of up to 256 queries costs one GEMM pass
    // (~3.7 TB/s over the row-major matrix, ~3 TB/s of binary16 over the nomination image) plus
    // ~0.8 ms for the sample, the per-query selects and the finish kernels.
    const double row_bytes = static_cast<double>(ix->n_rows) * ix->dim * (ix->dtype == RLR_F16 ? 2.0 : 4.0);
    // the single-query scan streams the 8-bit copy / the binary16 image when those are switched on
    const double scan_bytes = static_cast<double>(ix->n_rows) * ix->dim *
                              (scan_over_q8(ix) ? 1.0 : (scan_over_image(ix) || ix->dtype == RLR_F16) ? 2.0 : 4.0);
    const double t_single = 60e-6 + scan_bytes / 6.2e12;
    const bool image = ix->image_enabled && ix->d_image;
    // (over the image, batches of <= 128 queries take the resident-query kernel: ~5 TB/s of binary16)
    const double pass = image ? static_cast<double>(ix->n_rows) * ix->dim * 2.0 / (nq <= 128 ? 5.0e12 : 3.0e12)
                              : row_bytes / 3.7e12;
    const bool multi = nq <= 8 && ix->dtype == RLR_F32 && ix->pitch16 % 64 == 0 && ix->pitch16 / 64 <= 4 &&
                       ix->pitch16 * 4 == ix->dim && !image;
    if (multi) // one VALU pass for all of them + the per-query selects over the materialised scores
        return nq * t_single > 0.3e-3 + row_bytes / 5.5e12 + nq * (static_cast<double>(ix->n_rows) * 12.0 / 3.0e12);
    const double t_batch = 0.8e-3 + pass * ((nq + 255) / 256);
    return nq * t_single > t_batch;
}

// Runs queries [q0, q0+nq) (already staged in c->d_query) through the GEMM nomination pipeline
// and leaves k packed results per query in d_out; h_status[q] != 0 marks queries the caller must
// re-run through the single-query pipeline.
int32_t run_batched(rlr_index *ix, Ctx *c, uint32_t q0, uint32_t nq, const SearchPlan &p, uint64_t *d_out,
                    std::vector<uint32_t> &h_status, uint64_t *h_res_out)
{
    hipStream_t s = c->stream;
    const uint32_t n = static_cast<uint32_t>(ix->n_rows);
    const uint32_t fin_cap = batch_finish_capacity();
    const uint32_t n_qblocks = (nq + 255) / 256;
    // 2..8 queries over f32 rows: one VALU pass over the rows for all of them (scan_multi_kernel) instead of the
    // matrix-core pipeline -- about the cost of a single scan, scores in wavefront order (the tight f32 band)
    const bool use_multi = nq <= 8 && ix->dtype == RLR_F32 && ix->pitch16 % 64 == 0 && ix->pitch16 / 64 <= 4 &&
                           ix->pitch16 * 4 == ix->dim && !(ix->image_enabled && ix->d_image) && !env_is_one("RLR_NO_MULTI_SCAN");
    const float eps_nom = use_multi ? 0.5f * p.two_eps : nomination_eps(ix->dim, ix->dtype) * p.scale;
    const float two_eps = 2.0f * eps_nom;
    // Sample rows [0, S): the floor for the rest of the corpus is the sample's rank-th score, and S is large enough that
    // the expected number of later rows above it (rank * N / S) stays well inside the per-query candidate capacity.
    // rank = k needs no check afterwards (k sample rows sit at or above the floor) but a sample of k * N * 2.5 / capacity
    // rows -- so large that a quarter of it used to be materialised and the rest run as a separate "bootstrap" launch with
    // a per-query sort to tighten the floor (0.2 ms of a 4.4 ms batch of 256, 1.2 of 13.9 at 1024 x 308).  rank = k / 4
    // gives the same floor from the quarter alone: the count of rows above it spreads more (relative deviation
    // 1 / sqrt(rank): 3.3 k +- 0.65 k at rank 25 against a capacity of 8192), and the finish hands a query back when its
    // k-th nominated score lies under the floor's rank (batch_band_kernel; essentially never: it takes fewer than k rows
    // where ~3 k are expected).  RLR_BATCH_RANK_DIV=1 restores rank = k with the bootstrap.
    static const uint32_t rank_div = [] {
        const char *v = getenv("RLR_BATCH_RANK_DIV");
        const long d = v ? strtol(v, nullptr, 10) : 4;
        return static_cast<uint32_t>(d >= 1 && d <= 64 ? d : 4);
    }();
    uint32_t rank = std::max<uint32_t>(std::min<uint32_t>(p.k, 16), p.k / rank_div);
    uint64_t S = (static_cast<uint64_t>(rank) * n * 5 / 2 + fin_cap - 1) / fin_cap;
    S = std::max<uint64_t>(S, std::min<uint64_t>(n, 65536));
    S = (S + 255) / 256 * 256;
    if (S * 2 >= n || use_multi) {
        S = n;
        rank = p.k; // the sample is the corpus: its k-th score is the k-th score
    } else {
        // fewer than k rows above the floor <=> the sample holds `rank` of the corpus' best k - 1 rows, a Poisson(k S / N)
        // count: keep the rank six deviations above that mean (it matters when the 65 536-row minimum makes the sample a
        // large part of a small corpus; at 10 M rows the mean is 0.8)
        const double mean_in_sample = static_cast<double>(p.k) * static_cast<double>(S) / n;
        const uint32_t safe = static_cast<uint32_t>(std::ceil(mean_in_sample + 6.0 * std::sqrt(mean_in_sample) + 6.0));
        rank = std::min<uint32_t>(p.k, std::max(rank, safe));
        static const uint32_t forced = getenv("RLR_BATCH_RANK_FORCE") ? static_cast<uint32_t>(atoi(getenv("RLR_BATCH_RANK_FORCE"))) : 0;
        if (forced) // tests: a rank low enough that queries ARE handed back (too few candidates, or a band below the floor)
            rank = std::min<uint32_t>(p.k, forced);
    }
    // Bootstrap: materialising and radix-selecting S rows per query costs 4 passes over nq x S floats
    // (2.4 GB for 1024 queries x 587 k rows).  When S is large, only a quarter of it (S1) goes that way;
    // its threshold filters the rows [S1, S) in the GEMM epilogue (about 3 k extra candidates), one sort
    // per query tightens the threshold to the k-th score of all S rows, and the main pass starts from the
    // same threshold the full-size sample would have given.
    const uint64_t S2 = S;
    uint64_t S1 = std::max<uint64_t>(65536, (S2 / 4 + 255) / 256 * 256);
    const bool bootstrap = rank_div == 1 && S2 < n && S1 * 2 <= S2 && !getenv("RLR_BATCH_NO_BOOTSTRAP");
    if (bootstrap)
        S = S1;
    const uint64_t s_stride = (S + 3) / 4 * 4;

    // workspace
    const uint64_t qfrag_bytes = static_cast<uint64_t>(n_qblocks) * 256 * ix->dim * 2;
    if (c->qfrag_cap < qfrag_bytes) {
        if (c->d_qfrag) (void)hipFree(c->d_qfrag);
        c->d_qfrag = nullptr;
        c->qfrag_cap = 0;
        RLR_HIP(rlr::dev_malloc(&c->d_qfrag, qfrag_bytes));
        c->qfrag_cap = qfrag_bytes;
    }
    if (c->bq_cap < nq) {
        (void)hipFree(c->d_tau);
        (void)hipFree(c->d_bstate);
        (void)hipFree(c->d_bhist);
        (void)hipFree(c->d_bstatus);
        if (c->h_batch) (void)hipHostFree(c->h_batch);
        c->h_batch = nullptr;
        c->d_tau = nullptr;
        c->d_bstate = nullptr;
        c->d_bhist = nullptr;
        c->d_bstatus = nullptr;
        c->bq_cap = 0;
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_tau), static_cast<size_t>(nq) * sizeof(float)));
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_bstate), static_cast<size_t>(nq) * sizeof(SelectState)));
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_bhist), static_cast<size_t>(nq) * 2 * kHistBins * sizeof(uint32_t)));
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_bstatus), static_cast<size_t>(nq) * sizeof(uint32_t)));
        RLR_HIP(hipHostMalloc(&c->h_batch, static_cast<size_t>(nq) * (sizeof(SelectState) + sizeof(uint32_t)), hipHostMallocDefault));
        c->bq_cap = nq;
    }
    if (!c->d_gsync)
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_gsync), 256 * sizeof(uint32_t)));
    RLR_TRY(grow(&c->d_bcand, &c->bcand_cap, static_cast<uint64_t>(nq) * fin_cap));
    RLR_TRY(grow(&c->d_sample, &c->sample_cap, static_cast<uint64_t>(nq) * s_stride));

    SelectState *h_st = static_cast<SelectState *>(c->h_batch);
    uint32_t *h_stat = reinterpret_cast<uint32_t *>(h_st + c->bq_cap);
    for (uint32_t i = 0; i < nq; ++i) {
        std::memset(&h_st[i], 0, sizeof(SelectState));
        h_st[i].k = std::min<uint32_t>(rank, static_cast<uint32_t>(S));
        h_st[i].cap = fin_cap;
    }
    const float *dq = c->d_query + static_cast<size_t>(q0) * ix->q_pitch;
    const bool timed = ix->profiling;
    RLR_HIP(hipMemcpyAsync(c->d_bstate, h_st, nq * sizeof(SelectState), hipMemcpyHostToDevice, s));
    RLR_HIP(hipMemsetAsync(c->d_bhist, 0, static_cast<size_t>(nq) * 2 * kHistBins * sizeof(uint32_t), s));
    RLR_HIP(hipMemsetAsync(c->d_bstatus, 0xFF, static_cast<size_t>(nq) * sizeof(uint32_t), s));
    if (timed) RLR_HIP(hipEventRecord(c->bev[0], s));
    const bool use_image = ix->image_enabled && ix->d_image && gemm_image_usable(ix->dim);
    const void *image = use_image ? ix->d_image : nullptr;
    // the image stores k in natural order (like binary16 rows); only the direct f32-row loads permute it
    if (use_multi) {
        ScanArgs sa;
        sa.rows = ix->d_rows;
        sa.query = dq;
        sa.scores = c->d_sample;
        sa.hist = nullptr;
        sa.n_rows = n;
        sa.dim = ix->dim;
        sa.pitch16 = ix->pitch16;
        sa.dtype = ix->dtype;
        sa.n_cu = ix->n_cu;
        sa.variant = ix->scan_variant;
        hipError_t e = hipSuccess;
        if (!launch_scan_multi(sa, ix->q_pitch, nq, s_stride, s, &e))
            return fail(RLR_E_INTERNAL, "multi-query scan refused a shape its gate accepted");
        RLR_HIP(e);
    } else {
        RLR_HIP(launch_prep_queries(dq, nq, ix->q_pitch, ix->dim, use_image ? static_cast<int>(RLR_F16) : ix->dtype, c->d_qfrag, s));
        // 1. nominated scores of the sample rows, materialised
        RLR_HIP(launch_gemm_nominate(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, 0, static_cast<uint32_t>(S), c->d_qfrag, nq,
                                     nullptr, nullptr, 0, nullptr, c->d_sample, s_stride, image, s));
    }
    if (timed) RLR_HIP(hipEventRecord(c->bev[1], s));
    // 2. per-query k-th score of the sample -> threshold; the sample's own candidates
    RLR_HIP(launch_batch_select(c->d_sample, static_cast<uint32_t>(S), s_stride, nq, c->d_bhist, c->d_bstate, two_eps,
                                c->d_tau, c->d_bcand, fin_cap, ix->n_cu, s));
    if (timed) RLR_HIP(hipEventRecord(c->bev[2], s));
    uint32_t rest_begin = static_cast<uint32_t>(S);
    if (bootstrap) {
        // 2b. rows [S1, S2) against the small sample's threshold, then tighten it
        RLR_HIP(launch_gemm_nominate(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, static_cast<uint32_t>(S1),
                                     static_cast<uint32_t>(S2), c->d_qfrag, nq, c->d_tau, c->d_bcand, fin_cap, c->d_bstate,
                                     nullptr, 0, image, s, c->d_gsync));
        RLR_HIP(launch_batch_tighten(c->d_bcand, fin_cap, c->d_bstate, nq, p.k, two_eps, c->d_tau, s));
        rest_begin = static_cast<uint32_t>(S2);
    }
    // 3. the rest of the corpus, filtered in the GEMM epilogue
    if (timed) RLR_HIP(hipEventRecord(c->bev[4], s));
    RLR_HIP(launch_gemm_nominate(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, rest_begin, n, c->d_qfrag, nq,
                                 c->d_tau, c->d_bcand, fin_cap, c->d_bstate, nullptr, 0, image, s, c->d_gsync));
    if (timed) RLR_HIP(hipEventRecord(c->bev[3], s));
    // 4. per-query finish: band, reference-order re-score, order, emit
    RLR_HIP(launch_batch_finish(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, dq, ix->q_pitch, nq, c->d_bcand, fin_cap,
                                c->d_bstate, p.k, two_eps, d_out, c->d_bstatus, s));
    if (timed) RLR_HIP(hipEventRecord(c->ev[3], s));
    RLR_HIP(hipMemcpyAsync(h_stat, c->d_bstatus, nq * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (h_res_out) // the batch's results ride on the same synchronisation (queries handed back are fetched again by the caller)
        RLR_HIP(hipMemcpyAsync(h_res_out, d_out, static_cast<size_t>(nq) * p.k * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    RLR_HIP(hipStreamSynchronize(s));
    h_status.assign(h_stat, h_stat + nq);
    uint64_t fallbacks = 0;
    for (uint32_t v : h_status)
        fallbacks += v != 0;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        ix->prof.n_batches += 1;
        ix->prof.n_batch_queries += nq;
        ix->prof.n_batch_fallbacks += fallbacks;
        if (!use_image && !use_multi && nq >= 16 && ix->dtype == RLR_F32 && gemm_image_usable(ix->dim))
            ix->prof.n_batches_without_image += 1; // (an image would have halved the bytes this batch streamed)
        if (timed) {
            float prep = 0, sel = 0, g2 = 0, fin = 0;
            (void)hipEventElapsedTime(&prep, c->bev[0], c->bev[1]);
            (void)hipEventElapsedTime(&sel, c->bev[1], c->bev[2]);
            (void)hipEventElapsedTime(&g2, c->bev[2], c->bev[3]);
            (void)hipEventElapsedTime(&fin, c->bev[3], c->ev[3]);
            ix->prof.batch_gemm_ms += prep + g2; // prep is ~us; both GEMM launches are in here
            ix->prof.batch_other_ms += sel + fin;
            const uint64_t opb = (ix->dtype == RLR_F16 || use_image) ? 2 : 4;
            ix->prof.batch_gemm_bytes += static_cast<uint64_t>(n) * ix->dim * opb;
            ix->prof.batch_gemm_flops += 2.0 * nq * static_cast<double>(n) * ix->dim;
            if (!use_multi && rest_begin < n) {
                float mainp = 0;
                (void)hipEventElapsedTime(&mainp, c->bev[4], c->bev[3]);
                ix->prof.batch_main_ms += mainp;
                ix->prof.batch_main_bytes += static_cast<uint64_t>(n - rest_begin) * ix->dim * opb;
                ix->prof.batch_main_flops += 2.0 * nq * static_cast<double>(n - rest_begin) * ix->dim;
            }
        }
    }
    return RLR_OK;
}

// Runs nq queries.  Packed results (k per query) end up in d_out_user when given (device-resident
// variant), else in the context buffer AND in pinned host memory (*h_results, nq x k u64) -- one
// H2D (queries), one D2H (results + per-query candidate counts) and one stream synchronisation
// per call on the common path.
int32_t run_search(rlr_index *ix, Ctx *c, const float *queries, uint32_t nq, uint32_t k_req, float guard_eps,
                   uint64_t *d_out_user, SearchPlan *plan_out, const uint64_t **h_results)
{
    const uint32_t n = static_cast<uint32_t>(ix->n_rows);
    SearchPlan p;
    p.k = std::min<uint32_t>(k_req, n);
    p.scale = band_scale(ix, queries, nq);
    const float eps = (guard_eps >= 0.0f ? guard_eps : rlr_default_guard_eps(ix->dim)) * p.scale;
    p.two_eps = 2.0f * eps;
    p.two_eps_img = image_two_eps(ix, eps);
    p.cap = kLdsSortCap;
    *plan_out = p;
    if (h_results)
        *h_results = nullptr;
    if (p.k == 0 || nq == 0)
        return RLR_OK;
    RLR_TRY(ctx_prepare(ix, c, nq, p));
    const size_t n_res = static_cast<size_t>(nq) * p.k;
    uint64_t *d_out = d_out_user ? d_out_user : c->d_out;
    uint64_t *d_meta = c->d_out + n_res; // per-query candidate counts, right behind the context's results

    // stage queries (zero padded to the row pitch)
    const size_t q_bytes = static_cast<size_t>(nq) * ix->q_pitch * sizeof(float);
    const size_t res_bytes = (n_res + nq) * sizeof(uint64_t);
    RLR_TRY(pin_reserve(c, q_bytes + res_bytes));
    float *h_q = static_cast<float *>(c->h_pin);
    uint64_t *h_res = reinterpret_cast<uint64_t *>(static_cast<char *>(c->h_pin) + q_bytes);
    uint64_t *h_meta = h_res + n_res;
    if (ix->q_pitch != ix->dim)
        std::memset(h_q, 0, q_bytes);
    for (uint32_t q = 0; q < nq; ++q)
        std::memcpy(h_q + static_cast<size_t>(q) * ix->q_pitch, queries + static_cast<size_t>(q) * ix->dim,
                    ix->dim * sizeof(float));
    stage_query_norms(ix, c, queries, nq);
    hipStream_t s = c->stream;
    c->hist_dirty = true; // cleared when every enqueued pipeline has run to its histogram-clearing stage
    const bool batched = batch_eligible(ix, nq, p.k);
    c->h_q_kq = nullptr;
    if (batched) // (the matrix-core pipeline reads the queries from device memory)
        RLR_HIP(upload_queries(c, h_q, q_bytes, s));
    else
        RLR_HIP(stage_queries_for_scans(ix, c, h_q, q_bytes, s));

    const bool timed = ix->profiling;
    double scan_ms = 0, select_ms = 0, rescore_ms = 0, total_ms = 0;
    uint64_t n_cand_total = 0, n_retry = 0;
    auto fetch_results = [&](size_t first, size_t count) -> int32_t { // packed results -> pinned host
        if (!d_out_user && count)
            RLR_HIP(hipMemcpyAsync(h_res + first, c->d_out + first, count * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        return RLR_OK;
    };

    if (batched) {
        // matrix-core path in runs of <= kBatchMaxQueries; queries it hands back (overflow, or
        // fewer than k finite candidates) go through the single-query pipeline below.
        std::vector<uint32_t> redo;
        std::vector<uint32_t> status;
        for (uint32_t q0 = 0; q0 < nq; q0 += kBatchMaxQueries) {
            const uint32_t m = std::min(kBatchMaxQueries, nq - q0);
            RLR_TRY(run_batched(ix, c, q0, m, p, d_out + static_cast<size_t>(q0) * p.k, status,
                                d_out_user ? nullptr : h_res + static_cast<size_t>(q0) * p.k));
            for (uint32_t i = 0; i < m; ++i)
                if (status[i] != 0)
                    redo.push_back(q0 + i);
        }
        for (uint32_t q : redo) {
            RLR_HIP(enqueue_query(ix, c, q, p, d_out + static_cast<size_t>(q) * p.k, d_meta + q, false));
            RLR_HIP(hipMemcpyAsync(h_meta + q, d_meta + q, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
            RLR_HIP(hipStreamSynchronize(s));
            const uint32_t nc = static_cast<uint32_t>(h_meta[q]);
            if (nc > p.cap || nc > kLdsSortCap) {
                n_retry++;
                RLR_TRY(big_query(ix, c, q, p, nc, /*rescan=*/false, d_out + static_cast<size_t>(q) * p.k));
            }
        }
        if (!redo.empty()) { // (their results changed after the batch's own copy)
            RLR_TRY(fetch_results(0, n_res));
            RLR_HIP(hipStreamSynchronize(s));
        }
        c->hist_dirty = false;
        if (h_results)
            *h_results = h_res;
        std::lock_guard<std::mutex> lk(ix->mu);
        ix->prof.n_searches += nq;
        ix->prof.n_retries += n_retry;
        return RLR_OK;
    }

    // Host-bound results: the last kernel of each pipeline writes its k packed results and the candidate
    // count straight into the pinned (device-mapped) host buffer -- no D2H copy operation on the stream.
    const bool host_direct = !d_out_user;
    uint64_t *q_out = host_direct ? h_res : d_out;
    uint64_t *q_meta = h_meta; // the candidate count always goes straight to the host (the overflow check needs it)
    if (wait_mode() != kWaitBlock)
        for (uint32_t q = 0; q < nq; ++q)
            h_meta[q] = kMetaPending; // each pipeline's last store replaces it (sort_emit.h): what the wait below polls
    if (!timed || nq == 1) {
        // (with profiling on, a single query's four events are read after the one final sync)
        for (uint32_t q = 0; q < nq; ++q)
            RLR_HIP(enqueue_query(ix, c, q, p, q_out + static_cast<size_t>(q) * p.k, q_meta + q, timed));
    } else {
        // one query at a time so the four events can be read back per query
        for (uint32_t q = 0; q < nq; ++q) {
            RLR_HIP(enqueue_query(ix, c, q, p, q_out + static_cast<size_t>(q) * p.k, q_meta + q, true));
            RLR_HIP(hipStreamSynchronize(s));
            float a = 0, b = 0, d = 0;
            RLR_HIP(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
            RLR_HIP(hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
            RLR_HIP(hipEventElapsedTime(&d, c->ev[2], c->ev[3]));
            scan_ms += a;
            select_ms += b;
            rescore_ms += d;
            total_ms += a + b + d;
        }
    }
    if (timed)
        RLR_HIP(hipStreamSynchronize(s)); // (the events are read below)
    else
        RLR_TRY(wait_results(h_meta, host_direct ? h_res : nullptr, nq, p.k, std::min<uint32_t>(p.cap, kLdsSortCap), s,
                             &c->wait_ema));
    RLR_TRY(check_hist_assert(c));
    if (timed && nq == 1) {
        float a = 0, b = 0, d = 0;
        RLR_HIP(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
        RLR_HIP(hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
        RLR_HIP(hipEventElapsedTime(&d, c->ev[2], c->ev[3]));
        scan_ms += a;
        select_ms += b;
        rescore_ms += d;
        total_ms += a + b + d;
    }
    // band overflow -> large-candidate path (rare: massive exact ties, or k > 4096)
    bool refetch = false;
    for (uint32_t q = 0; q < nq; ++q) {
        const uint32_t nc = static_cast<uint32_t>(h_meta[q]);
        n_cand_total += nc;
        if (nc > p.cap || nc > kLdsSortCap) {
            n_retry++;
            RLR_TRY(big_query(ix, c, q, p, nc, /*rescan=*/nq > 1, q_out + static_cast<size_t>(q) * p.k));
            refetch = !host_direct;
        }
    }
    if (refetch)
        RLR_HIP(hipStreamSynchronize(s));
    c->hist_dirty = false;
    if (h_results)
        *h_results = h_res;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        ix->prof.n_searches += nq;
        ix->prof.n_candidates += n_cand_total;
        ix->prof.n_retries += n_retry;
        if (timed) {
            ix->prof.n_scan_launches += nq;
            ix->prof.scan_ms += scan_ms;
            ix->prof.select_ms += select_ms;
            ix->prof.rescore_ms += rescore_ms;
            ix->prof.total_ms += total_ms;
            ix->prof.scan_bytes += static_cast<uint64_t>(nq) * ix->n_rows * ix->dim *
                                   (scan_over_q8(ix) ? 1 : (ix->dtype == RLR_F16 || scan_over_image(ix)) ? 2 : 4);
        }
    }
    return RLR_OK;
}

int32_t upload_list(rlr_index *ix, Ctx *c, const uint64_t *rows, uint32_t n, uint64_t bound = ~0ull)
{
    if (bound == ~0ull)
        bound = ix->n_rows;
    if (c->list_cap < n || !c->d_list) {
        if (c->d_list) (void)hipFree(c->d_list);
        if (c->d_vals) (void)hipFree(c->d_vals);
        c->d_list = nullptr;
        c->d_vals = nullptr;
        c->list_cap = 0;
        const uint32_t cap = std::max<uint32_t>(next_pow2(n), 1024);
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_list), static_cast<size_t>(cap) * sizeof(uint32_t)));
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_vals), static_cast<size_t>(cap) * sizeof(float)));
        c->list_cap = cap;
    }
    RLR_TRY(pin_reserve(c, static_cast<size_t>(n) * 8 + 64));
    uint32_t *h = static_cast<uint32_t *>(c->h_pin);
    for (uint32_t i = 0; i < n; ++i) {
        if (rows[i] >= bound)
            return fail(RLR_E_RANGE, "row %llu out of range (%llu rows)", static_cast<unsigned long long>(rows[i]),
                        static_cast<unsigned long long>(bound));
        h[i] = static_cast<uint32_t>(rows[i]);
    }
    RLR_HIP(hipMemcpyAsync(c->d_list, h, static_cast<size_t>(n) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    return RLR_OK;
}

} // namespace

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" {

int32_t rlr_version(void)
{
    return RLR_VERSION;
}

int32_t rlr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

const char *rlr_last_error(void)
{
    return g_err;
}

float rlr_default_guard_eps(uint32_t dim)
{
    // |seq_sum - tree_sum| <= (dim + reduction depth) * 2^-24 * sum|x_i y_i| <= that * |x||y|
    // for unit-norm operands; 64 covers the 12-deep lane chain + 6 DPP levels with margin,
    // and the final 1/16 absorbs norms that are 1 +- a few ulp.
    return (static_cast<float>(dim) + 64.0f) * 5.9604645e-8f * 1.0625f;
}

int32_t rlr_index_create(uint32_t dim, int32_t dtype, int32_t device_id, rlr_index **out)
{
    if (!out)
        return fail(RLR_E_INVALID, "out is null");
    *out = nullptr;
    if (dim == 0 || dim > kMaxDim)
        return fail(RLR_E_INVALID, "dim must be in [1, %u]", kMaxDim);
    if (dtype != RLR_F32 && dtype != RLR_F16)
        return fail(RLR_E_INVALID, "unknown dtype %d", dtype);
    int n_dev = rlr_device_count();
    if (n_dev <= 0)
        return fail(RLR_E_NO_DEVICE, "no HIP device is visible (this library has no CPU path)");
    if (device_id < 0 || device_id >= n_dev)
        return fail(RLR_E_NO_DEVICE, "device %d out of range (%d visible)", device_id, n_dev);
    rlr_index *ix = new (std::nothrow) rlr_index();
    if (!ix)
        return fail(RLR_E_OOM, "host allocation failed");
    ix->dim = dim;
    ix->dtype = dtype;
    ix->device = device_id;
    const uint32_t elem = dtype == RLR_F16 ? 2 : 4;
    ix->pitch16 = (dim * elem + 15) / 16;
    ix->q_pitch = ix->pitch16 * (16 / elem);
    hipDeviceProp_t prop;
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&prop, device_id) != hipSuccess) {
        delete ix;
        return fail(RLR_E_HIP, "cannot query device %d", device_id);
    }
    ix->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char *v = getenv("RLR_SCAN_VARIANT"))
        ix->scan_variant = static_cast<int>(strtol(v, nullptr, 0));
    if (const char *v = getenv("RLR_TAIL"))
        ix->fused_tail = v[0] != '0' ? 1 : 0;
    if (const char *v = getenv("RLR_TAIL_DIRECT_MAX"))
        ix->tail_direct_max = static_cast<uint32_t>(std::min<unsigned long>(strtoul(v, nullptr, 0), 4096));
    if (const char *v = getenv("RLR_HYBRID_FETCH"))
        ix->hybrid_fetch_full = !strcmp(v, "full");
    if (const char *v = getenv("RLR_BATCH_MIN"))
        ix->batch_min = static_cast<uint32_t>(strtoul(v, nullptr, 0));
    if (const char *v = getenv("RLR_MAX_CONTEXTS"))
        ix->ctx_cap = static_cast<int>(std::min<long>(std::max<long>(strtol(v, nullptr, 0), 1), 64));
    *out = ix;
    return RLR_OK;
}

int32_t rlr_index_destroy(rlr_index *ix)
{
    if (!ix)
        return RLR_OK;
    (void)hipSetDevice(ix->device);
    (void)hipDeviceSynchronize();
    for (Ctx *c : ix->free_ctx)
        ctx_free(c);
    rows_free(&ix->rows_block);
    ix->d_rows = nullptr;
    if (ix->d_q8) (void)hipFree(ix->d_q8);
    if (ix->d_q8_scale) (void)hipFree(ix->d_q8_scale);
    if (ix->d_q8_stats) (void)hipFree(ix->d_q8_stats);
    if (ix->d_image)
        (void)hipFree(ix->d_image);
    delete ix;
    return RLR_OK;
}

int32_t rlr_index_info(const rlr_index *ix, uint64_t *n_rows, uint32_t *dim, int32_t *dtype, int32_t *device_id)
{
    RLR_TRY(check_handle(ix));
    if (n_rows) *n_rows = ix->n_rows;
    if (dim) *dim = ix->dim;
    if (dtype) *dtype = ix->dtype;
    if (device_id) *device_id = ix->device;
    return RLR_OK;
}

int32_t rlr_index_reserve(rlr_index *ix, uint64_t n_rows)
{
    RLR_TRY(check_handle(ix));
    RLR_TRY(use_device(ix));
    return ensure_rows(ix, n_rows);
}

int32_t rlr_index_upload(rlr_index *ix, const float *rows, uint64_t n_rows, int32_t normalize_on_device)
{
    RLR_TRY(check_handle(ix));
    if (n_rows && !rows)
        return fail(RLR_E_INVALID, "rows is null");
    RLR_TRY(use_device(ix));
    ix->n_rows = 0;
    ix->max_row_sumsq = 1.0f;
    RLR_TRY(ensure_rows(ix, n_rows));
    RLR_TRY(ingest(ix, rows, n_rows, 0, normalize_on_device));
    ix->n_rows = n_rows;
    return sync_image(ix, 0);
}

int32_t rlr_index_append(rlr_index *ix, const float *rows, uint64_t n_rows, int32_t normalize_on_device,
                         uint64_t *first_row_out)
{
    RLR_TRY(check_handle(ix));
    if (n_rows && !rows)
        return fail(RLR_E_INVALID, "rows is null");
    RLR_TRY(use_device(ix));
    const uint64_t first = ix->n_rows;
    RLR_TRY(ensure_rows(ix, first + n_rows));
    RLR_TRY(ingest(ix, rows, n_rows, first, normalize_on_device));
    ix->n_rows = first + n_rows;
    if (first_row_out)
        *first_row_out = first;
    return sync_image(ix, first);
}

int32_t rlr_index_delete_rows(rlr_index *ix, const uint64_t *rows, uint64_t n)
{
    RLR_TRY(check_handle(ix));
    if (n == 0)
        return RLR_OK;
    if (!rows)
        return fail(RLR_E_INVALID, "rows is null");
    RLR_TRY(use_device(ix));
    const uint64_t N = ix->n_rows;
    std::vector<uint8_t> dead(N, 0);
    for (uint64_t i = 0; i < n; ++i) {
        if (rows[i] >= N)
            return fail(RLR_E_RANGE, "row %llu out of range (index holds %llu rows)",
                        static_cast<unsigned long long>(rows[i]), static_cast<unsigned long long>(N));
        dead[rows[i]] = 1;
    }
    uint64_t first_dead = 0;
    while (first_dead < N && !dead[first_dead])
        ++first_dead;
    std::vector<uint32_t> keep;
    keep.reserve(N - first_dead);
    for (uint64_t r = first_dead; r < N; ++r)
        if (!dead[r])
            keep.push_back(static_cast<uint32_t>(r));
    // stable in-place compaction through a bounce buffer: destination rows
    // [first_dead + i0, first_dead + i1) only overwrite rows below every source still to move.
    const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / row_bytes(ix));
    void *d_bounce = nullptr;
    uint32_t *d_keep = nullptr;
    const uint64_t cr = std::min<uint64_t>(chunk, std::max<size_t>(keep.size(), 1));
    RLR_HIP(rlr::dev_malloc(&d_bounce, cr * row_bytes(ix)));
    hipError_t e = rlr::dev_malloc(reinterpret_cast<void **>(&d_keep), cr * sizeof(uint32_t));
    if (e != hipSuccess) {
        (void)hipFree(d_bounce);
        return fail(RLR_E_OOM, "compaction buffer allocation failed");
    }
    int32_t st = RLR_OK;
    for (uint64_t i0 = 0; i0 < keep.size() && st == RLR_OK; i0 += cr) {
        const uint64_t m = std::min<uint64_t>(cr, keep.size() - i0);
        e = hipMemcpy(d_keep, keep.data() + i0, m * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = launch_compact_rows(ix->d_rows, d_bounce, ix->pitch16, d_keep, static_cast<uint32_t>(m), nullptr);
        if (e == hipSuccess)
            e = hipMemcpyAsync(static_cast<char *>(ix->d_rows) + (first_dead + i0) * row_bytes(ix), d_bounce,
                               m * row_bytes(ix), hipMemcpyDeviceToDevice, nullptr);
        if (e == hipSuccess)
            e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess)
            st = fail(RLR_E_HIP, "row compaction failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_bounce);
    (void)hipFree(d_keep);
    if (st == RLR_OK) {
        ix->n_rows = first_dead + keep.size();
        st = sync_image(ix, first_dead);
    }
    return st;
}

int32_t rlr_index_enable_batch_image(rlr_index *ix, int32_t enable)
{
    RLR_TRY(check_handle(ix));
    RLR_TRY(use_device(ix));
    const bool want_image = (enable & 3) != 0, want_q8 = (enable & 4) != 0;
    if (!want_q8 && ix->q8_enabled) {
        ix->q8_enabled = false;
        if (ix->d_q8) (void)hipFree(ix->d_q8);
        if (ix->d_q8_scale) (void)hipFree(ix->d_q8_scale);
        ix->d_q8 = nullptr;
        ix->d_q8_scale = nullptr;
        ix->q8_cap_rows = 0;
    }
    if (!want_image) {
        ix->image_enabled = false;
        ix->image_scan = false;
        if (ix->d_image)
            (void)hipFree(ix->d_image);
        ix->d_image = nullptr;
        ix->image_cap = 0;
    }
    if (want_q8 && !ix->q8_enabled) {
        if (ix->dim % 16 != 0 || ix->dim > 2048)
            return fail(RLR_E_INVALID, "the 8-bit nomination copy needs dim %% 16 == 0, dim <= 2048 (dim = %u)", ix->dim);
        ix->q8_enabled = true;
        RLR_TRY(sync_q8(ix, 0));
    }
    if (want_image) {
        if (ix->dim % 64 != 0)
            return fail(RLR_E_INVALID, "the nomination image needs dim %% 64 == 0 (dim = %u)", ix->dim);
        ix->image_scan = (enable & 2) != 0;
        if (!(ix->image_enabled && ix->d_image)) {
            ix->image_enabled = true;
            const bool q8 = ix->q8_enabled;
            ix->q8_enabled = false; // the 8-bit copy is current: rebuild the image only
            const int32_t st = sync_image(ix, 0);
            ix->q8_enabled = q8;
            RLR_TRY(st);
        }
    }
    return RLR_OK;
}

int32_t rlr_index_fill_synthetic(rlr_index *ix, uint64_t n_rows, uint64_t row0, uint64_t seed, uint32_t n_clusters)
{
    RLR_TRY(check_handle(ix));
    RLR_TRY(use_device(ix));
    ix->n_rows = 0;
    RLR_TRY(ensure_rows(ix, n_rows));
    const uint64_t chunk = 1ull << 20;
    float *d_norm = nullptr;
    RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&d_norm), std::min(chunk, std::max<uint64_t>(n_rows, 1)) * sizeof(float)));
    int32_t st = RLR_OK;
    // bit 30: the last n_rows / 100 rows repeat the first n_rows / 100 (exact duplicates)
    const uint64_t n_dup = (n_clusters & 0x40000000u) ? n_rows / 100 : 0;
    const uint64_t seam = n_rows - n_dup; // rows [seam, n_rows) are generator rows [row0, row0 + n_dup) again
    n_clusters &= ~0x40000000u;
    for (uint64_t r0 = 0; r0 < n_rows && st == RLR_OK;) {
        uint64_t m = std::min(chunk, n_rows - r0);
        if (r0 < seam)
            m = std::min(m, seam - r0); // (a launch never straddles the seam)
        const uint64_t src = r0 < seam ? row0 + r0 : row0 + (r0 - seam);
        hipError_t e = launch_synth(static_cast<char *>(ix->d_rows) + r0 * row_bytes(ix), ix->pitch16, ix->dim, ix->dtype,
                                    src, static_cast<uint32_t>(m), seed, n_clusters, d_norm, nullptr);
        if (e == hipSuccess)
            e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess)
            st = fail(RLR_E_HIP, "synthetic fill failed: %s", hipGetErrorString(e));
        r0 += m;
    }
    (void)hipFree(d_norm);
    if (st == RLR_OK) {
        ix->n_rows = n_rows;
        st = sync_image(ix, 0);
    }
    return st;
}

int32_t rlr_search_topk(rlr_index *ix, const float *queries, uint32_t n_queries, uint32_t k, float guard_eps,
                        uint64_t *rows_out, float *cos_out, uint32_t *n_out)
{
    RLR_TRY(check_handle(ix));
    if (n_queries && (!queries || !n_out))
        return fail(RLR_E_INVALID, "queries / n_out is null");
    if (n_queries && k && (!rows_out || !cos_out))
        return fail(RLR_E_INVALID, "output buffers are null");
    RLR_TRY(use_device(ix));
    if (ix->n_rows == 0 || k == 0) {
        for (uint32_t q = 0; q < n_queries; ++q)
            n_out[q] = 0;
        return RLR_OK;
    }
    CtxLease lease(ix);
    RLR_TRY(ctx_acquire(ix, &lease.c));
    Ctx *c = lease.c;
    SearchPlan p;
    const uint64_t *h = nullptr;
    RLR_TRY(run_search(ix, c, queries, n_queries, k, guard_eps, nullptr, &p, &h));
    if (!h)
        return fail(RLR_E_INTERNAL, "search produced no result buffer");
    for (uint32_t q = 0; q < n_queries; ++q) {
        n_out[q] = p.k;
        // unpack_result, written without branches so that the loop vectorises (a batch of 1024 x 308 results is 315 k of them)
        const uint64_t *__restrict__ src = h + static_cast<size_t>(q) * p.k;
        uint64_t *__restrict__ ro = rows_out + static_cast<size_t>(q) * k;
        uint32_t *__restrict__ co = reinterpret_cast<uint32_t *>(cos_out + static_cast<size_t>(q) * k);
        for (uint32_t i = 0; i < p.k; ++i) {
            const uint64_t w = src[i];
            const uint32_t key = static_cast<uint32_t>(w >> 32);
            const uint32_t neg = static_cast<uint32_t>(static_cast<int32_t>(key) >> 31);        // all ones: key of a value >= +0
            const uint32_t bits = key ^ (0xFFFFFFFFu ^ (neg & 0x7FFFFFFFu));                     // key_score(): & 0x7FFFFFFF or ~
            co[i] = key == 0u ? 0x7FC00000u : bits;
            ro[i] = 0xFFFFFFFFu - static_cast<uint32_t>(w);
        }
    }
    return RLR_OK;
}

int32_t rlr_search_topk_device(rlr_index *ix, const float *queries, uint32_t n_queries, uint32_t k, float guard_eps,
                               void *d_packed_out, void *stream)
{
    RLR_TRY(check_handle(ix));
    if (n_queries && k && (!queries || !d_packed_out))
        return fail(RLR_E_INVALID, "queries / d_packed_out is null");
    RLR_TRY(use_device(ix));
    if (n_queries == 0 || k == 0)
        return RLR_OK;
    hipStream_t user = static_cast<hipStream_t>(stream);
    if (ix->n_rows == 0) {
        RLR_HIP(hipMemsetAsync(d_packed_out, 0, static_cast<size_t>(n_queries) * k * sizeof(uint64_t), user));
        return RLR_OK;
    }
    CtxLease lease(ix);
    RLR_TRY(ctx_acquire(ix, &lease.c));
    Ctx *c = lease.c;
    SearchPlan p;
    if (k <= ix->n_rows) {
        RLR_TRY(run_search(ix, c, queries, n_queries, k, guard_eps, static_cast<uint64_t *>(d_packed_out), &p, nullptr));
    } else {
        // fewer rows than k: produce the n_rows results, then spread them into k-strided slots
        const uint64_t *h_res = nullptr;
        RLR_TRY(run_search(ix, c, queries, n_queries, k, guard_eps, nullptr, &p, &h_res));
        RLR_HIP(hipMemsetAsync(d_packed_out, 0, static_cast<size_t>(n_queries) * k * sizeof(uint64_t), c->stream));
        RLR_HIP(hipMemcpy2DAsync(d_packed_out, static_cast<size_t>(k) * 8, h_res, static_cast<size_t>(p.k) * 8,
                                 static_cast<size_t>(p.k) * 8, n_queries, hipMemcpyHostToDevice, c->stream));
        RLR_HIP(hipStreamSynchronize(c->stream));
    }
    // run_search has synchronised the context stream, so the results are complete; a later
    // enqueue on `user` is ordered after them.
    (void)user;
    return RLR_OK;
}

// Asynchronous flavour for the sharded step: begin() enqueues the pipelines and makes `stream` wait for
// them, so the caller can queue its collective and merge behind the scan without a host round trip; end()
// joins, reports queries whose guard band overflowed (their slots in d_packed_out are not valid: the caller
// re-runs the step through rlr_search_topk_device) and returns the context.
int32_t rlr_search_topk_device_begin(rlr_index *ix, const float *queries, uint32_t n_queries, uint32_t k, float guard_eps,
                                     void *d_packed_out, void *stream, void **ticket_out)
{
    RLR_TRY(check_handle(ix));
    if (!ticket_out)
        return fail(RLR_E_INVALID, "ticket_out is null");
    *ticket_out = nullptr;
    if (n_queries && k && (!queries || !d_packed_out))
        return fail(RLR_E_INVALID, "queries / d_packed_out is null");
    RLR_TRY(use_device(ix));
    // anything but the plain single-query pipelines runs synchronously (ticket stays null)
    const bool timed = ix->profiling; // one query: its four events are read in end()
    if (n_queries == 0 || k == 0 || ix->n_rows == 0 || k > ix->n_rows || (timed && n_queries > 1) ||
        batch_eligible(ix, n_queries, k))
        return rlr_search_topk_device(ix, queries, n_queries, k, guard_eps, d_packed_out, stream);
    Ctx *c = nullptr;
    RLR_TRY(ctx_acquire(ix, &c));
    CtxLease lease(ix);
    lease.c = c; // released on every error path below
    SearchPlan p;
    p.k = k;
    p.scale = band_scale(ix, queries, n_queries);
    p.two_eps = 2.0f * (guard_eps >= 0.0f ? guard_eps : rlr_default_guard_eps(ix->dim)) * p.scale;
    p.two_eps_img = image_two_eps(ix, p.two_eps * 0.5f);
    p.cap = kLdsSortCap;
    RLR_TRY(ctx_prepare(ix, c, n_queries, p));
    const size_t q_bytes = static_cast<size_t>(n_queries) * ix->q_pitch * sizeof(float);
    RLR_TRY(pin_reserve(c, q_bytes + (static_cast<size_t>(n_queries) * k + n_queries) * sizeof(uint64_t)));
    float *h_q = static_cast<float *>(c->h_pin);
    uint64_t *h_meta = reinterpret_cast<uint64_t *>(static_cast<char *>(c->h_pin) + q_bytes) + static_cast<size_t>(n_queries) * k;
    if (ix->q_pitch != ix->dim)
        std::memset(h_q, 0, q_bytes);
    for (uint32_t q = 0; q < n_queries; ++q)
        std::memcpy(h_q + static_cast<size_t>(q) * ix->q_pitch, queries + static_cast<size_t>(q) * ix->dim,
                    ix->dim * sizeof(float));
    stage_query_norms(ix, c, queries, n_queries);
    // The pipelines go on the CALLER's stream: whatever it queues next (all-gather, merge) is ordered behind
    // them by the stream itself.  (A cross-stream event wait was measured first: +20 us per step.)  The
    // context's own stream is idle -- every earlier use of this context ended with a synchronisation.
    hipStream_t own = c->stream;
    c->stream = static_cast<hipStream_t>(stream);
    hipStream_t s = c->stream;
    c->hist_dirty = true;
    if (wait_mode() != kWaitBlock)
        for (uint32_t q = 0; q < n_queries; ++q)
            h_meta[q] = kMetaPending; // (what _end polls instead of the stream's completion signal)
    hipError_t e = stage_queries_for_scans(ix, c, h_q, q_bytes, s);
    uint64_t *out = static_cast<uint64_t *>(d_packed_out);
    for (uint32_t q = 0; q < n_queries && e == hipSuccess; ++q)
        e = enqueue_query(ix, c, q, p, out + static_cast<size_t>(q) * k, h_meta + q, timed);
    c->stream = own;
    if (e != hipSuccess) {
        // pipelines already enqueued on the caller's stream still use this context's buffers: wait for them before
        // the lease hands the context back to the pool (ctx_acquire's recovery only knows the context's own stream)
        (void)hipStreamSynchronize(s);
        return fail(RLR_E_HIP, "enqueue on the caller's stream failed: %s", hipGetErrorString(e));
    }
    c->pending_stream = s;
    c->pending_timed = timed;
    c->pending_q = n_queries;
    c->pending_meta = h_meta;
    c->pending_k = k;
    lease.c = nullptr; // the ticket owns the context until end()
    *ticket_out = c;
    return RLR_OK;
}

int32_t rlr_search_topk_device_end(rlr_index *ix, void *ticket, uint32_t *n_overflow_out)
{
    RLR_TRY(check_handle(ix));
    if (n_overflow_out)
        *n_overflow_out = 0;
    if (!ticket)
        return RLR_OK; // begin() ran synchronously
    Ctx *c = static_cast<Ctx *>(ticket);
    CtxLease lease(ix);
    lease.c = c;
    RLR_TRY(use_device(ix));
    if (c->pending_timed)
        RLR_HIP(hipStreamSynchronize(c->pending_stream)); // (the events are read below)
    else // usually already there: whatever the caller queued behind the pipelines and waited for ran after them
        RLR_TRY(wait_results(c->pending_meta, nullptr, c->pending_q, c->pending_k, kLdsSortCap, c->pending_stream, nullptr));
    RLR_TRY(check_hist_assert(c));
    uint32_t over = 0;
    uint64_t n_cand = 0;
    for (uint32_t q = 0; q < c->pending_q; ++q) {
        const uint32_t nc = static_cast<uint32_t>(c->pending_meta[q]);
        n_cand += nc;
        over += nc > kLdsSortCap;
    }
    c->hist_dirty = false;
    float t_scan = 0, t_sel = 0, t_res = 0;
    if (c->pending_timed) {
        RLR_HIP(hipEventElapsedTime(&t_scan, c->ev[0], c->ev[1]));
        RLR_HIP(hipEventElapsedTime(&t_sel, c->ev[1], c->ev[2]));
        RLR_HIP(hipEventElapsedTime(&t_res, c->ev[2], c->ev[3]));
    }
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        ix->prof.n_searches += c->pending_q;
        ix->prof.n_candidates += n_cand;
        ix->prof.n_retries += over;
        if (c->pending_timed) {
            ix->prof.n_scan_launches += c->pending_q;
            ix->prof.scan_ms += t_scan;
            ix->prof.select_ms += t_sel;
            ix->prof.rescore_ms += t_res;
            ix->prof.total_ms += t_scan + t_sel + t_res;
            ix->prof.scan_bytes += static_cast<uint64_t>(c->pending_q) * ix->n_rows * ix->dim *
                                   (scan_over_q8(ix) ? 1 : (ix->dtype == RLR_F16 || scan_over_image(ix)) ? 2 : 4);
        }
    }
    c->pending_q = 0;
    c->pending_timed = false;
    if (n_overflow_out)
        *n_overflow_out = over;
    return RLR_OK;
}

int32_t rlr_merge_topk(int32_t device_id, const void *d_gathered, uint32_t world, uint32_t n_queries, uint32_t k,
                       const uint64_t *bases, uint64_t *rows_out, float *cos_out, uint32_t *n_out, void *stream)
{
    if (n_queries == 0 || k == 0)
        return RLR_OK;
    if (!d_gathered || !bases || !rows_out || !cos_out || !n_out)
        return fail(RLR_E_INVALID, "null argument");
    if (world == 0 || world > 16)
        return fail(RLR_E_INVALID, "world size %u not in [1, 16]", world);
    if (static_cast<uint64_t>(world) * k > 8192)
        return fail(RLR_E_INVALID, "world * k = %llu exceeds the 8192-entry merge", static_cast<unsigned long long>(world) * k);
    RLR_HIP(hipSetDevice(device_id));
    // results are written straight into pinned, device-mapped host memory: no D2H copy
    thread_local void *h_buf = nullptr;
    thread_local size_t h_cap = 0;
    const size_t nk = static_cast<size_t>(n_queries) * k;
    const size_t need = nk * (sizeof(uint64_t) + sizeof(float)) + n_queries * (sizeof(uint32_t) + sizeof(uint64_t)) + 64;
    if (h_cap < need) {
        if (h_buf)
            (void)hipHostFree(h_buf);
        h_buf = nullptr;
        h_cap = 0;
        RLR_HIP(hipHostMalloc(&h_buf, std::max<size_t>(need, 1 << 16), hipHostMallocDefault));
        h_cap = std::max<size_t>(need, 1 << 16);
    }
    uint64_t *h_rows = static_cast<uint64_t *>(h_buf);
    uint64_t *h_flag = h_rows + nk;
    float *h_cos = reinterpret_cast<float *>(h_flag + n_queries);
    uint32_t *h_n = reinterpret_cast<uint32_t *>(h_cos + nk);
    rlr::MergeBases mb;
    for (uint32_t r = 0; r < 16; ++r)
        mb.base[r] = r < world ? bases[r] : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (uint32_t q = 0; q < n_queries; ++q)
        h_flag[q] = kMetaPending;
    hipLaunchKernelGGL(rlr::merge_topk_kernel, dim3(n_queries), dim3(1024), 0, s, static_cast<const uint64_t *>(d_gathered),
                       world, n_queries, k, mb, h_rows, h_cos, h_n, h_flag);
    RLR_HIP(hipGetLastError());
    // (this wait usually spans the scans queued in front of the merge on the same stream)
    thread_local WaitEma merge_wait;
    const volatile uint64_t *vf = h_flag;
    const volatile uint64_t *vr = h_rows;
    const volatile uint32_t *vc = reinterpret_cast<const volatile uint32_t *>(h_cos);
    uint32_t verified = n_queries;
    RLR_TRY(wait_polling(
        [&]() {
            while (verified > 0) {
                const uint32_t q = verified - 1;
                const uint64_t m = vf[q];
                if (m == kMetaPending)
                    return false;
                uint32_t chk = 0;
                for (uint32_t i = 0; i < k; ++i)
                    chk += result_chk_term(vr[static_cast<size_t>(q) * k + i], i) +
                           result_chk_term(vc[static_cast<size_t>(q) * k + i], i + k);
                if (chk != static_cast<uint32_t>(m >> 32))
                    return false;
                verified--;
            }
            return true;
        },
        s, &merge_wait, world * 65536u + n_queries));
    std::memcpy(rows_out, h_rows, nk * sizeof(uint64_t));
    std::memcpy(cos_out, h_cos, nk * sizeof(float));
    std::memcpy(n_out, h_n, n_queries * sizeof(uint32_t));
    return RLR_OK;
}

uint64_t rlr_pack_result(float score, uint32_t row)
{
    return pack_result(score, row);
}

void rlr_unpack_result(uint64_t packed, float *score, uint32_t *row)
{
    float s;
    uint32_t r;
    unpack_result(packed, &s, &r);
    if (score) *score = s;
    if (row) *row = r;
}

int32_t rlr_score_rows(rlr_index *ix, const float *query, const uint64_t *rows, uint32_t n, float *cos_out)
{
    RLR_TRY(check_handle(ix));
    if (n == 0)
        return RLR_OK;
    if (!query || !rows || !cos_out)
        return fail(RLR_E_INVALID, "null argument");
    RLR_TRY(use_device(ix));
    CtxLease lease(ix);
    RLR_TRY(ctx_acquire(ix, &lease.c));
    Ctx *c = lease.c;
    RLR_TRY(upload_list(ix, c, rows, n));
    if (c->q_cap < 1) {
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_query), static_cast<size_t>(ix->q_pitch) * sizeof(float)));
        RLR_HIP(rlr::dev_malloc(reinterpret_cast<void **>(&c->d_state), sizeof(SelectState)));
        RLR_HIP(hipMemsetAsync(c->d_state, 0, sizeof(SelectState), c->stream)); // (the fused tail's counters, as in ctx_prepare)
        c->q_cap = 1;
    }
    RLR_HIP(hipMemcpyAsync(c->d_query, query, ix->dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
    RLR_HIP(launch_score_rows(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, c->d_query, c->d_list, n, c->d_vals,
                              c->stream));
    RLR_HIP(hipMemcpyAsync(cos_out, c->d_vals, static_cast<size_t>(n) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RLR_HIP(hipStreamSynchronize(c->stream));
    return RLR_OK;
}

int32_t rlr_fetch_rows(rlr_index *ix, const uint64_t *rows, uint32_t n, float *out)
{
    RLR_TRY(check_handle(ix));
    if (n == 0)
        return RLR_OK;
    if (!rows || !out)
        return fail(RLR_E_INVALID, "null argument");
    RLR_TRY(use_device(ix));
    CtxLease lease(ix);
    RLR_TRY(ctx_acquire(ix, &lease.c));
    Ctx *c = lease.c;
    RLR_TRY(upload_list(ix, c, rows, n));
    RLR_TRY(grow(&c->d_pool, &c->pool_cap, static_cast<uint64_t>(n) * ix->dim));
    RLR_HIP(launch_gather_f32(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, c->d_list, n, c->d_pool, c->stream));
    RLR_HIP(hipMemcpyAsync(out, c->d_pool, static_cast<size_t>(n) * ix->dim * sizeof(float), hipMemcpyDeviceToHost,
                           c->stream));
    RLR_HIP(hipStreamSynchronize(c->stream));
    return RLR_OK;
}

// profile bookkeeping of an MMR call: c->ev[0] .. c->ev[1] bracket gather + Gram + greedy on the context's stream
static void note_mmr(rlr_index *ix, Ctx *c, uint32_t n_queries)
{
    float ms = 0;
    (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->prof.n_mmr += n_queries;
    ix->prof.mmr_ms += ms;
}

// one pool of up to 4096 candidates; d_matrix != null: pool_rows are slots of that staged matrix (see mmr_batch_impl)
static int32_t mmr_single_impl(rlr_index *ix, const uint64_t *pool_rows, const float *pool_scores, uint32_t P, uint32_t k,
                               float lambda, uint32_t *order_out, float *mmr_out, uint32_t *n_out, const void *d_matrix,
                               uint64_t n_matrix)
{
    RLR_TRY(check_handle(ix));
    if (!n_out)
        return fail(RLR_E_INVALID, "n_out is null");
    *n_out = 0;
    if (P == 0)
        return RLR_OK; // `if candidates.is_empty() { return vec![] }`
    if (!pool_rows || !pool_scores || !order_out)
        return fail(RLR_E_INVALID, "null argument");
    if (P > 4096)
        return fail(RLR_E_INVALID, "pool of %u exceeds the supported 4096 candidates", P);
    RLR_TRY(use_device(ix));
    CtxLease lease(ix);
    RLR_TRY(ctx_acquire(ix, &lease.c));
    Ctx *c = lease.c;
    hipStream_t s = c->stream;
    RLR_TRY(upload_list(ix, c, pool_rows, P, d_matrix ? n_matrix : ~0ull));
    // pool (P x dim) followed by gram (P x P), scores (P), order (P), mmr (P), n (1)
    const uint64_t floats = static_cast<uint64_t>(P) * ix->dim + static_cast<uint64_t>(P) * P + 3ull * P + 4;
    RLR_TRY(grow(&c->d_pool, &c->pool_cap, floats));
    float *d_pool = c->d_pool;
    float *d_gram = d_pool + static_cast<uint64_t>(P) * ix->dim;
    float *d_sc = d_gram + static_cast<uint64_t>(P) * P;
    uint32_t *d_order = reinterpret_cast<uint32_t *>(d_sc + P);
    float *d_mmr = d_sc + 2ull * P;
    uint32_t *d_n = reinterpret_cast<uint32_t *>(d_sc + 3ull * P);
    const bool timed = ix->profiling;
    RLR_HIP(hipMemcpyAsync(d_sc, pool_scores, static_cast<size_t>(P) * sizeof(float), hipMemcpyHostToDevice, s));
    if (timed) RLR_HIP(hipEventRecord(c->ev[0], s));
    RLR_HIP(launch_gram_rows(d_matrix ? d_matrix : ix->d_rows, ix->pitch16, ix->dim, ix->dtype, c->d_list, P, d_gram, 1, s));
    RLR_HIP(launch_mmr_greedy(d_gram, d_sc, P, k, lambda, d_order, d_mmr, d_n, nullptr, 1, s));
    if (timed) RLR_HIP(hipEventRecord(c->ev[1], s));
    // one D2H: order | mmr | n are contiguous
    RLR_TRY(pin_reserve(c, (2ull * P + 4) * 4));
    RLR_HIP(hipMemcpyAsync(c->h_pin, d_order, (2ull * P + 1) * 4, hipMemcpyDeviceToHost, s));
    RLR_HIP(hipStreamSynchronize(s));
    if (timed)
        note_mmr(ix, c, 1);
    const uint32_t *h_order = static_cast<const uint32_t *>(c->h_pin);
    const float *h_mmr = reinterpret_cast<const float *>(h_order + P);
    const uint32_t n_sel = h_order[2 * P];
    for (uint32_t i = 0; i < n_sel; ++i) {
        order_out[i] = h_order[i];
        if (mmr_out)
            mmr_out[i] = h_mmr[i];
    }
    *n_out = n_sel;
    return RLR_OK;
}

int32_t rlr_mmr_select(rlr_index *ix, const uint64_t *pool_rows, const float *pool_scores, uint32_t P, uint32_t k,
                       float lambda, uint32_t *order_out, float *mmr_out, uint32_t *n_out)
{
    return mmr_single_impl(ix, pool_rows, pool_scores, P, k, lambda, order_out, mmr_out, n_out, nullptr, 0);
}

int32_t rlr_search_diverse(rlr_index *ix, const float *query, uint32_t pool, uint32_t k, float lambda, float w_embedding,
                           float w_lexical, float guard_eps, uint64_t *rows_out, float *cos_out, float *score_out,
                           uint32_t *n_out, int32_t *fallback)
{
    RLR_TRY(check_handle(ix));
    if (!n_out || !fallback)
        return fail(RLR_E_INVALID, "n_out / fallback is null");
    *n_out = 0;
    *fallback = 0;
    if (ix->n_rows == 0 || pool == 0)
        return RLR_OK;
    if (!query || !rows_out || !cos_out || !score_out)
        return fail(RLR_E_INVALID, "null argument");
    const uint32_t n = static_cast<uint32_t>(ix->n_rows);
    const uint32_t need = std::min<uint32_t>(n, pool);
    const uint32_t fetch = static_cast<uint32_t>(std::min<uint64_t>(n, static_cast<uint64_t>(need) + 8));
    if (need > rlr::kPoolMax || !(w_embedding > 0.0f) || !(w_lexical >= 0.0f) || !std::isfinite(w_lexical)) {
        *fallback = 1; // outside what the fused kernels cover: the caller's two-call path handles it
        return RLR_OK;
    }
    RLR_TRY(use_device(ix));
    CtxLease lease(ix);
    RLR_TRY(ctx_acquire(ix, &lease.c));
    Ctx *c = lease.c;
    hipStream_t s = c->stream;
    SearchPlan p;
    p.k = fetch;
    p.scale = band_scale(ix, query, 1);
    const float eps = (guard_eps >= 0.0f ? guard_eps : rlr_default_guard_eps(ix->dim)) * p.scale;
    p.two_eps = 2.0f * eps;
    p.two_eps_img = image_two_eps(ix, eps);
    p.cap = kLdsSortCap;
    RLR_TRY(ctx_prepare(ix, c, 1, p));
    const uint32_t P = need;
    const uint32_t k_cap = std::max<uint32_t>(std::min<uint32_t>(std::max<uint32_t>(k, 1u), P), 1u);
    // workspace: pool P x dim | gram P x P | combined P | cos P | order P | mmr P | n_sel, info[2]
    const uint64_t floats = static_cast<uint64_t>(P) * ix->dim + static_cast<uint64_t>(P) * P + 4ull * P + 8;
    RLR_TRY(grow(&c->d_pool, &c->pool_cap, floats));
    if (c->list_cap < P || !c->d_list) {
        const uint64_t zero = 0;
        RLR_TRY(upload_list(ix, c, &zero, 1)); // (allocates the list for >= 1024 rows)
    }
    float *d_pool = c->d_pool;
    float *d_gram = d_pool + static_cast<uint64_t>(P) * ix->dim;
    float *d_comb = d_gram + static_cast<uint64_t>(P) * P;
    float *d_cos = d_comb + P;
    uint32_t *d_order = reinterpret_cast<uint32_t *>(d_cos + P);
    float *d_mmr = d_cos + 2ull * P;
    uint32_t *d_nsel = reinterpret_cast<uint32_t *>(d_cos + 3ull * P);
    uint32_t *d_info = d_nsel + 1;
    const size_t q_bytes = static_cast<size_t>(ix->q_pitch) * sizeof(float);
    const size_t out_words = 4ull * k_cap + 4;
    RLR_TRY(pin_reserve(c, q_bytes + out_words * 4 + 64));
    float *h_q = static_cast<float *>(c->h_pin);
    uint32_t *h_out = reinterpret_cast<uint32_t *>(static_cast<char *>(c->h_pin) + q_bytes);
    h_out[4 * k_cap + 3] = kBlockPending; // (the greedy kernel's last store replaces it: what the wait below polls)
    std::memset(h_q, 0, q_bytes);
    std::memcpy(h_q, query, ix->dim * sizeof(float));
    stage_query_norms(ix, c, query, 1);
    c->hist_dirty = true;
    const bool timed = ix->profiling;
    RLR_HIP(stage_queries_for_scans(ix, c, h_q, q_bytes, s));
    uint64_t *d_meta = c->d_out + fetch;
    static const bool two_launches = getenv("RLR_POOL_AFTER_SORT") != nullptr; // A/B: sort_emit, then the pool from its output
    const bool from_candidates = fetch <= 512 && !two_launches;                // (the band of a larger fetch rarely fits 1024)
    const PoolArgs pa{fetch, need, n, w_embedding, w_lexical, c->d_list, d_comb, d_cos, d_info};
    bool pool_done = false; // (the fused tail's finish builds the pool itself: one launch and ~13 us of config 2's chain less)
    RLR_HIP(enqueue_query(ix, c, 0, p, c->d_out, d_meta, timed, /*emit=*/!from_candidates, from_candidates ? &pa : nullptr,
                          &pool_done));
    if (timed) RLR_HIP(hipEventRecord(c->bev[0], s));
    if (!pool_done) {
        if (from_candidates)
            hipLaunchKernelGGL(rlr::pool_prepare_kernel<true>, dim3(1), dim3(1024), 0, s, c->d_packed, c->d_state, pa);
        else
            hipLaunchKernelGGL(rlr::pool_prepare_kernel<false>, dim3(1), dim3(1024), 0, s, c->d_out, c->d_state, pa);
        RLR_HIP(hipGetLastError());
    }
    RLR_HIP(launch_gram_rows(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, c->d_list, P, d_gram, 1, s));
    rlr::MmrEmit emit; // the greedy kernel writes the picks into the pinned block itself
    emit.list = c->d_list;
    emit.comb = d_comb;
    emit.cosv = d_cos;
    emit.info = d_info;
    emit.k_cap = k_cap;
    emit.h_out = h_out;
    RLR_HIP(launch_mmr_greedy(d_gram, d_comb, P, k, lambda, d_order, d_mmr, d_nsel, d_info, 1, s, &emit));
    if (timed) RLR_HIP(hipEventRecord(c->bev[1], s));
    if (timed)
        RLR_HIP(hipStreamSynchronize(s)); // (the events are read below)
    else
        RLR_TRY(wait_block(h_out, k_cap, s, &c->wait_ema));
    RLR_TRY(check_hist_assert(c));
    c->hist_dirty = false;
    const uint32_t n_sel = h_out[4 * k_cap], status = h_out[4 * k_cap + 1];
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        ix->prof.n_searches += 1;
        if (timed) {
            float a = 0, b = 0, d = 0, m = 0;
            (void)hipEventElapsedTime(&a, c->ev[0], c->ev[1]);
            (void)hipEventElapsedTime(&b, c->ev[1], c->ev[2]);
            (void)hipEventElapsedTime(&d, c->ev[2], c->ev[3]);
            (void)hipEventElapsedTime(&m, c->bev[0], c->bev[1]);
            ix->prof.n_scan_launches += 1;
            ix->prof.scan_ms += a;
            ix->prof.select_ms += b;
            ix->prof.rescore_ms += d;
            ix->prof.total_ms += a + b + d + m;
            ix->prof.scan_bytes += ix->n_rows * ix->dim * (scan_over_q8(ix) ? 1 : (ix->dtype == RLR_F16 || scan_over_image(ix)) ? 2 : 4);
            ix->prof.n_mmr += 1;
            ix->prof.mmr_ms += m;
        }
    }
    if (status != 0) {
        *fallback = static_cast<int32_t>(status);
        return RLR_OK;
    }
    for (uint32_t i = 0; i < n_sel; ++i) {
        rows_out[i] = h_out[i];
        cos_out[i] = __builtin_bit_cast(float, h_out[k_cap + i]);
        score_out[i] = __builtin_bit_cast(float, h_out[2 * k_cap + i]);
    }
    *n_out = n_sel;
    return RLR_OK;
}

// where the lexical pairs of a hybrid search come from
struct HybridLexSrc {
    const uint64_t *h_rows = nullptr; // host pairs: ascending unique rows inside the index ...
    const float *h_scores = nullptr;
    uint32_t n_host = 0;
    float max_lex = 1.1920929e-07f;
    const rlr::LexPending *dev = nullptr; // ... or a BM25 call's result still on the device (null + n_host 0: no pairs)
};

} // extern "C"

namespace rlr {

// A hybrid search between its two enqueues: the scan / select / re-score / sort part is on the stream (it does not
// depend on the lexical pairs, only on an upper bound of their number), the blend and everything behind it follows in
// search_hybrid_finish.  In between the caller enqueues the BM25 kernels on their own stream: the device runs them
// beside the scan, and the host's launch calls for both overlap the scan instead of preceding it.
struct HybridTicket {
    rlr_index *ix;
    CtxLease lease;
    uint32_t n, need, fetch, n_lex_bound, k, k_cap;
    float lambda, w_e, w_l;
    int32_t diversify;
    bool timed;
    size_t q_bytes, lex_bytes_cap;
    explicit HybridTicket(rlr_index *i) : ix(i), lease(i) {}
};

static int32_t hybrid_begin_impl(rlr_index *ix, const float *query, uint32_t need_in, uint32_t k, float lambda, int32_t diversify,
                                 float w_embedding, float w_lexical, uint32_t n_lex_bound, float guard_eps, HybridTicket **out,
                                 int32_t *fallback, int32_t (*behind_scan)(void *, const LexSink *) = nullptr,
                                 void *behind_scan_arg = nullptr)
{
    *out = nullptr;
    *fallback = 0;
    RLR_TRY(check_handle(ix));
    if (!query)
        return fail(RLR_E_INVALID, "null argument");
    const uint32_t n = static_cast<uint32_t>(ix->n_rows);
    const uint32_t need = std::min<uint32_t>(n, need_in);
    const uint64_t fetch_full = std::min<uint64_t>(n, static_cast<uint64_t>(need) + n_lex_bound + 8);
    const uint64_t fetch64 = ix->hybrid_fetch_full ? fetch_full : std::min<uint64_t>(fetch_full, static_cast<uint64_t>(need) + kHybridFetchMargin);
    if (n == 0 || need == 0 || need > kPoolMax || n_lex_bound > kHybridLexMax || fetch_full + n_lex_bound > kHybridSlots ||
        !(w_embedding > 0.0f) || !std::isfinite(w_embedding) || !std::isfinite(w_lexical)) {
        *fallback = 1; // outside what the fused kernels cover: the caller's host path handles it
        return RLR_OK;
    }
    const uint32_t fetch = static_cast<uint32_t>(fetch64);
    RLR_TRY(use_device(ix));
    std::unique_ptr<HybridTicket> t(new (std::nothrow) HybridTicket(ix));
    if (!t)
        return fail(RLR_E_OOM, "host allocation failed");
    RLR_TRY(ctx_acquire(ix, &t->lease.c));
    Ctx *c = t->lease.c;
    hipStream_t s = c->stream;
    SearchPlan p;
    p.k = fetch;
    p.scale = band_scale(ix, query, 1);
    const float eps = (guard_eps >= 0.0f ? guard_eps : rlr_default_guard_eps(ix->dim)) * p.scale;
    p.two_eps = 2.0f * eps;
    p.two_eps_img = image_two_eps(ix, eps);
    p.cap = kLdsSortCap;
    p.unordered = true; // the blend orders fetched and lexical rows together: it needs the fetched SET and its minimum
    RLR_TRY(ctx_prepare(ix, c, 1, p));
    const uint32_t P = need;
    const uint32_t k_cap = diversify ? std::max<uint32_t>(std::min<uint32_t>(std::max<uint32_t>(k, 1u), P), 1u) : P;
    // workspace (4-byte words): gram P x P | combined P | cos P | lex P | order P | mmr P | n_sel, info[2], pad |
    //                           header[2] | lexical rows | scores | cosines (bound each) | candidate combined / cos / lex
    const uint64_t words = static_cast<uint64_t>(P) * P + 5ull * P + 8 + 2 + 3ull * n_lex_bound + 3ull * kHybridSlots;
    RLR_TRY(grow(&c->d_pool, &c->pool_cap, words));
    if (c->list_cap < P || !c->d_list) {
        const uint64_t zero = 0;
        RLR_TRY(upload_list(ix, c, &zero, 1)); // (allocates the list for >= 1024 rows)
    }
    t->n = n;
    t->need = need;
    t->fetch = fetch;
    t->n_lex_bound = n_lex_bound;
    t->k = k;
    t->k_cap = k_cap;
    t->lambda = lambda;
    t->w_e = w_embedding;
    t->w_l = w_lexical;
    t->diversify = diversify;
    t->timed = ix->profiling;
    t->q_bytes = static_cast<size_t>(ix->q_pitch) * sizeof(float);
    t->lex_bytes_cap = 8 + static_cast<size_t>(n_lex_bound) * 8; // header | rows | scores, one copy
    RLR_TRY(pin_reserve(c, t->q_bytes + t->lex_bytes_cap + (4ull * k_cap + 4) * 4 + 64));
    float *h_q = static_cast<float *>(c->h_pin);
    std::memset(h_q, 0, t->q_bytes);
    std::memcpy(h_q, query, ix->dim * sizeof(float));
    stage_query_norms(ix, c, query, 1);
    c->hist_dirty = true;
    // from here on work may be queued on `s`: an error exit drains it before the lease hands the context back
    struct Drain {
        hipStream_t s;
        bool armed = true;
        ~Drain()
        {
            if (armed)
                (void)hipStreamSynchronize(s);
        }
    } drain{s};
    RLR_HIP(stage_queries_for_scans(ix, c, h_q, t->q_bytes, s));
    uint64_t *d_meta = c->d_out + fetch;
    // the scan first; then whatever the caller runs beside it (the BM25 chain on its own stream: about as long as scan +
    // select + re-score + sort, so it must not wait for the host to have launched those -- it used to start 39 us behind the
    // scan and was the critical path by as much); then the four launches that wait for the scan anyway
    RLR_HIP(enqueue_query_scan(ix, c, 0, t->timed));
    if (behind_scan) {
        // (the lexical side of the workspace as hybrid_finish_impl lays it out for pairs that stay on the device)
        uint32_t *d_nsel = reinterpret_cast<uint32_t *>(c->d_pool + static_cast<uint64_t>(P) * P + 5ull * P);
        LexSink sink;
        sink.d_header = d_nsel + 8;
        sink.d_rows = d_nsel + 10;
        sink.d_scores = reinterpret_cast<float *>(sink.d_rows + n_lex_bound);
        sink.n_bound = n_lex_bound;
        sink.n_index_rows = n;
        RLR_TRY(behind_scan(behind_scan_arg, &sink));
    }
    RLR_HIP(enqueue_query_rest(ix, c, 0, p, c->d_out, d_meta, t->timed));
    if (t->timed) RLR_HIP(hipEventRecord(c->bev[0], s));
    drain.armed = false;
    *out = t.release();
    return RLR_OK;
}

// always consumes the ticket
static int32_t hybrid_finish_impl(HybridTicket *ticket, const HybridLexSrc &src, uint64_t *rows_out, float *cos_out,
                                  float *score_out, float *lex_out, uint32_t *n_out, int32_t *fallback)
{
    std::unique_ptr<HybridTicket> t(ticket);
    rlr_index *ix = t->ix;
    Ctx *c = t->lease.c;
    hipStream_t s = c->stream;
    *n_out = 0;
    *fallback = 0;
    // whatever goes wrong from here on: the enqueued work may still be running on `s` when the context goes back
    struct Drain {
        hipStream_t s;
        bool armed = true;
        ~Drain()
        {
            if (armed)
                (void)hipStreamSynchronize(s);
        }
    } drain{s};
    const uint32_t n_lex = src.dev ? t->n_lex_bound : src.n_host;
    if (n_lex > t->n_lex_bound || !rows_out || !cos_out || !score_out || !lex_out)
        return fail(RLR_E_INVALID, "hybrid search: bad arguments");
    const uint32_t P = t->need, k_cap = t->k_cap, n = t->n;
    float *d_gram = c->d_pool;
    float *d_comb = d_gram + static_cast<uint64_t>(P) * P;
    float *d_cos = d_comb + P;
    float *d_lexv = d_cos + P;
    uint32_t *d_order = reinterpret_cast<uint32_t *>(d_lexv + P);
    float *d_mmr = d_lexv + 2ull * P;
    uint32_t *d_nsel = reinterpret_cast<uint32_t *>(d_lexv + 3ull * P);
    uint32_t *d_info = d_nsel + 1;
    HybridLexHeader *d_hdr = reinterpret_cast<HybridLexHeader *>(d_nsel + 8);
    uint32_t *d_lrow = d_nsel + 10;
    float *d_lscore = reinterpret_cast<float *>(d_lrow + n_lex);
    float *d_lcos = d_lscore + t->n_lex_bound;
    float *d_cand = d_lcos + t->n_lex_bound;
    uint32_t *h_lex = reinterpret_cast<uint32_t *>(static_cast<char *>(c->h_pin) + t->q_bytes);
    uint32_t *h_out = reinterpret_cast<uint32_t *>(static_cast<char *>(c->h_pin) + t->q_bytes + t->lex_bytes_cap);
    h_out[4 * k_cap + 3] = kBlockPending; // (the last kernel's last store replaces it: what the wait below polls)
    if (src.dev) { // the BM25 kernels ran beside the scan on their own stream: join, then unpack their result
        RLR_HIP(hipStreamWaitEvent(s, static_cast<hipEvent_t>(src.dev->ready), 0));
        if (!src.dev->unpacked) { // (begin's LexSink: the scoring stream has done it in front of `ready`)
            hipLaunchKernelGGL(lex_unpack_kernel, dim3(1), dim3(1024), 0, s, src.dev->d_packed, src.dev->d_count,
                               std::min(n_lex, src.dev->limit), n, d_lrow, d_lscore, d_hdr);
            RLR_HIP(hipGetLastError());
        }
    } else {
        for (uint32_t i = 0; i < n_lex; ++i)
            if (src.h_rows[i] >= n || (i && src.h_rows[i] <= src.h_rows[i - 1]))
                return fail(RLR_E_INVALID, "lex_rows must be ascending, unique and inside the index");
        h_lex[0] = n_lex;
        h_lex[1] = __builtin_bit_cast(uint32_t, src.max_lex);
        for (uint32_t i = 0; i < n_lex; ++i)
            h_lex[2 + i] = static_cast<uint32_t>(src.h_rows[i]);
        if (n_lex)
            std::memcpy(h_lex + 2 + n_lex, src.h_scores, static_cast<size_t>(n_lex) * sizeof(float));
        // header, rows, scores are adjacent on both sides
        RLR_HIP(hipMemcpyAsync(d_hdr, h_lex, 8 + static_cast<size_t>(n_lex) * 8, hipMemcpyHostToDevice, s));
    }
    RLR_HIP(launch_score_rows(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, c->d_query, d_lrow, n_lex, d_lcos, s,
                              src.dev ? &d_hdr->n_lex : nullptr, n));
    static const bool emit_split = getenv("RLR_HYBRID_EMIT") && !strcmp(getenv("RLR_HYBRID_EMIT"), "split"); // (A/B: its own launch)
    const bool pool_emits = !t->diversify && !emit_split;
    hipLaunchKernelGGL(hybrid_pool_kernel, dim3(1), dim3(1024), 0, s, c->d_out, t->fetch, t->need, n, t->w_e, t->w_l, d_lrow,
                       d_lscore, d_lcos, d_hdr, d_cand, c->d_list, d_comb, d_cos, d_lexv, d_info, k_cap,
                       pool_emits ? h_out : static_cast<uint32_t *>(nullptr));
    RLR_HIP(hipGetLastError());
    if (t->diversify) {
        RLR_HIP(launch_gram_rows(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, c->d_list, P, d_gram, 1, s));
        MmrEmit emit; // the greedy kernel writes the picks into the pinned block itself
        emit.list = c->d_list;
        emit.comb = d_comb;
        emit.cosv = d_cos;
        emit.lexv = d_lexv;
        emit.info = d_info;
        emit.k_cap = k_cap;
        emit.h_out = h_out;
        RLR_HIP(launch_mmr_greedy(d_gram, d_comb, P, t->k, t->lambda, d_order, d_mmr, d_nsel, d_info, 1, s, &emit));
    } else if (!pool_emits) {
        hipLaunchKernelGGL(hybrid_emit_kernel, dim3(1), dim3(256), 0, s, c->d_list, d_comb, d_cos, d_lexv, d_info, k_cap, h_out);
        RLR_HIP(hipGetLastError());
    }
    if (t->timed) RLR_HIP(hipEventRecord(c->bev[1], s));
    if (t->timed) {
        drain.armed = false;
        RLR_HIP(hipStreamSynchronize(s)); // (the events are read below)
    } else {
        RLR_TRY(wait_block(h_out, k_cap, s, &c->wait_ema)); // (a failure leaves the drain guard armed)
        drain.armed = false;
    }
    RLR_TRY(check_hist_assert(c));
    c->hist_dirty = false;
    const uint32_t n_sel = h_out[4 * k_cap], status = h_out[4 * k_cap + 1];
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        ix->prof.n_searches += 1;
        if (t->timed) {
            float a = 0, b = 0, d = 0, m = 0;
            (void)hipEventElapsedTime(&a, c->ev[0], c->ev[1]);
            (void)hipEventElapsedTime(&b, c->ev[1], c->ev[2]);
            (void)hipEventElapsedTime(&d, c->ev[2], c->ev[3]);
            (void)hipEventElapsedTime(&m, c->bev[0], c->bev[1]);
            ix->prof.n_scan_launches += 1;
            ix->prof.scan_ms += a;
            ix->prof.select_ms += b;
            ix->prof.rescore_ms += d;
            ix->prof.total_ms += a + b + d + m;
            ix->prof.scan_bytes += ix->n_rows * ix->dim * (scan_over_q8(ix) ? 1 : (ix->dtype == RLR_F16 || scan_over_image(ix)) ? 2 : 4);
            ix->prof.n_mmr += 1;
            ix->prof.mmr_ms += m;
        }
    }
    if (status != 0) {
        *fallback = static_cast<int32_t>(status);
        return RLR_OK;
    }
    for (uint32_t i = 0; i < n_sel; ++i) {
        rows_out[i] = h_out[i];
        cos_out[i] = __builtin_bit_cast(float, h_out[k_cap + i]);
        score_out[i] = __builtin_bit_cast(float, h_out[2 * k_cap + i]);
        lex_out[i] = __builtin_bit_cast(float, h_out[3 * k_cap + i]);
    }
    *n_out = n_sel;
    return RLR_OK;
}

void launch_lex_unpack(const uint64_t *d_packed, const uint32_t *d_count, uint32_t limit, const LexSink &sink, void *stream)
{
    // (tried: the event recorded by this launch itself, hipExtLaunchKernelGGL's stop event, instead of a marker packet behind
    // it -- the join on the search's stream resumed 3 us earlier under the profiler, nothing measurable without it)
    hipLaunchKernelGGL(lex_unpack_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), d_packed, d_count, limit,
                       sink.n_index_rows, sink.d_rows, sink.d_scores, static_cast<HybridLexHeader *>(sink.d_header));
}

int32_t search_hybrid_begin(rlr_index *ix, const float *query, uint32_t need, uint32_t k, float lambda, int32_t diversify,
                            float w_embedding, float w_lexical, uint32_t n_lex_bound, float guard_eps, HybridTicket **ticket,
                            int32_t *fallback, int32_t (*behind_scan)(void *, const LexSink *), void *behind_scan_arg)
{
    return hybrid_begin_impl(ix, query, need, k, lambda, diversify, w_embedding, w_lexical, n_lex_bound, guard_eps, ticket,
                             fallback, behind_scan, behind_scan_arg);
}

int32_t search_hybrid_finish(HybridTicket *ticket, const LexPending *lex, uint64_t *rows_out, float *cos_out, float *score_out,
                             float *lex_out, uint32_t *n_out, int32_t *fallback)
{
    HybridLexSrc src;
    if (lex && lex->limit)
        src.dev = lex; // else: no lexical pair at all -- the blend degenerates to w_e * cos + w_l * 0
    return hybrid_finish_impl(ticket, src, rows_out, cos_out, score_out, lex_out, n_out, fallback);
}

void search_hybrid_abort(HybridTicket *ticket)
{
    if (!ticket)
        return;
    (void)hipStreamSynchronize(ticket->lease.c->stream);
    delete ticket;
}

} // namespace rlr

extern "C" {

int32_t rlr_search_hybrid(rlr_index *ix, const float *query, uint32_t need, uint32_t k, float lambda, int32_t diversify,
                          float w_embedding, float w_lexical, const uint64_t *lex_rows, const float *lex_scores, uint32_t n_lex,
                          float max_lex, float guard_eps, uint64_t *rows_out, float *cos_out, float *score_out, float *lex_out,
                          uint32_t *n_out, int32_t *fallback)
{
    RLR_TRY(check_handle(ix));
    if (!n_out || !fallback)
        return fail(RLR_E_INVALID, "n_out / fallback is null");
    *n_out = 0;
    *fallback = 0;
    if (ix->n_rows == 0 || need == 0)
        return RLR_OK;
    if (!query || !rows_out || !cos_out || !score_out || !lex_out || (n_lex && (!lex_rows || !lex_scores)))
        return fail(RLR_E_INVALID, "null argument");
    rlr::HybridTicket *t = nullptr;
    RLR_TRY(rlr::hybrid_begin_impl(ix, query, need, k, lambda, diversify, w_embedding, w_lexical, n_lex, guard_eps, &t, fallback));
    if (*fallback)
        return RLR_OK;
    HybridLexSrc src;
    src.h_rows = lex_rows;
    src.h_scores = lex_scores;
    src.n_host = n_lex;
    src.max_lex = max_lex;
    return rlr::hybrid_finish_impl(t, src, rows_out, cos_out, score_out, lex_out, n_out, fallback);
}

// Batched MMR over P-strided pools.  The pool rows either live in the index (pool_rows != null:
// gathered to f32 here) or are already in device memory as n_queries x P x dim f32 values
// (d_values != null: the sharded path, after the winner-row exchange).
// d_matrix != null: pool_rows are slots of that matrix (n_matrix raw rows in the index' own dtype and pitch, e.g. the
// receive buffer of the cross-shard winner-row exchange) instead of rows of the index.
static int32_t mmr_batch_impl(rlr_index *ix, const uint64_t *pool_rows, const float *d_values, const float *pool_scores,
                              const uint32_t *pool_sizes, uint32_t n_queries, uint32_t P, uint32_t k, float lambda,
                              uint32_t *order_out, float *mmr_out, uint32_t *n_out, const void *d_matrix = nullptr,
                              uint64_t n_matrix = 0)
{
    RLR_TRY(check_handle(ix));
    if (n_queries == 0)
        return RLR_OK;
    if ((!pool_rows && !d_values) || !pool_scores || !pool_sizes || !order_out || !n_out)
        return fail(RLR_E_INVALID, "null argument");
    if (P == 0) {
        for (uint32_t q = 0; q < n_queries; ++q)
            n_out[q] = 0;
        return RLR_OK;
    }
    if (P > 1024)
        return fail(RLR_E_INVALID, "batched MMR supports pools of at most 1024 candidates (got %u)", P);
    RLR_TRY(use_device(ix));
    CtxLease lease(ix);
    RLR_TRY(ctx_acquire(ix, &lease.c));
    Ctx *c = lease.c;
    hipStream_t s = c->stream;
    // queries per pass: the workspace (gram m x P x P + scores / order / mmr) is capped at ~3 GB.  All queries of a
    // pass run their greedy chains concurrently, one wavefront each, so the more queries per pass the better the
    // chip is filled: at 64 per pass (the first version) 1024 pools took 16 rounds of gather + Gram + greedy + sync.
    const uint64_t per_query = static_cast<uint64_t>(P) * P + 3ull * P + 2;
    const uint32_t QC = static_cast<uint32_t>(std::max<uint64_t>(64, std::min<uint64_t>(4096, (3ull << 30) / 4 / std::max<uint64_t>(per_query, 1))));
    std::vector<uint64_t> rows_chunk;
    for (uint32_t q0 = 0; q0 < n_queries; q0 += QC) {
        const uint32_t m = std::min(QC, n_queries - q0);
        const uint32_t n_list = m * P;
        for (uint32_t q = 0; q < m; ++q)
            if (pool_sizes[q0 + q] > P)
                return fail(RLR_E_INVALID, "pool_sizes[%u] = %u exceeds P = %u", q0 + q, pool_sizes[q0 + q], P);
        // pinned staging layout: [row list (upload_list)] [scores] [sizes] [results]; reserve it all
        // BEFORE upload_list enqueues its copy so the buffer is never reallocated under a transfer
        const size_t list_bytes = static_cast<size_t>(n_list) * 8 + 64;
        const size_t in_bytes = static_cast<size_t>(n_list) * 4 + static_cast<size_t>(m + 4) * 4;
        const size_t out_bytes = (2ull * n_list + m) * 4;
        RLR_TRY(pin_reserve(c, list_bytes + in_bytes + out_bytes + 64));
        if (pool_rows) {
            // unused slots (j >= pool_sizes[q]) gather row 0: never read by the greedy kernel
            bool all_full = true;
            for (uint32_t q = 0; q < m && all_full; ++q)
                all_full = pool_sizes[q0 + q] == P;
            const uint64_t *src_rows = pool_rows + static_cast<size_t>(q0) * P;
            if (!all_full) { // (full pools are taken as they stand: no 8 n_list-byte fill + copy in front of every batch)
                rows_chunk.assign(n_list, 0);
                for (uint32_t q = 0; q < m; ++q)
                    std::memcpy(rows_chunk.data() + static_cast<size_t>(q) * P, pool_rows + static_cast<size_t>(q0 + q) * P,
                                pool_sizes[q0 + q] * sizeof(uint64_t));
                src_rows = rows_chunk.data();
            }
            RLR_TRY(upload_list(ix, c, src_rows, n_list, d_matrix ? n_matrix : ~0ull));
        }
        const uint64_t pool_floats = 0; // (the pool rows are read in place)
        const uint64_t floats = pool_floats + static_cast<uint64_t>(m) * P * P + 3ull * n_list + 2ull * m + 8;
        RLR_TRY(grow(&c->d_pool, &c->pool_cap, floats));
        float *d_pool = c->d_pool;
        float *d_gram = d_pool + pool_floats;
        float *d_sc = d_gram + static_cast<uint64_t>(m) * P * P;
        uint32_t *d_order = reinterpret_cast<uint32_t *>(d_sc + n_list);
        float *d_mmr = d_sc + 2ull * n_list;
        uint32_t *d_n = reinterpret_cast<uint32_t *>(d_sc + 3ull * n_list);
        uint32_t *d_sizes = d_n + m;
        char *h_base = static_cast<char *>(c->h_pin) + list_bytes;
        float *h_sc = reinterpret_cast<float *>(h_base);
        uint32_t *h_sizes = reinterpret_cast<uint32_t *>(h_sc + n_list);
        std::memcpy(h_sc, pool_scores + static_cast<size_t>(q0) * P, static_cast<size_t>(n_list) * sizeof(float));
        std::memcpy(h_sizes, pool_sizes + q0, m * sizeof(uint32_t));
        RLR_HIP(hipMemcpyAsync(d_sc, h_sc, static_cast<size_t>(n_list) * sizeof(float), hipMemcpyHostToDevice, s));
        RLR_HIP(hipMemcpyAsync(d_sizes, h_sizes, m * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        const bool timed = ix->profiling;
        if (timed) RLR_HIP(hipEventRecord(c->ev[0], s));
        if (pool_rows) // the Gram kernel reads the index rows through the list: no gathered copy
            RLR_HIP(launch_gram_rows(d_matrix ? d_matrix : ix->d_rows, ix->pitch16, ix->dim, ix->dtype, c->d_list, P, d_gram, m, s));
        else
            RLR_HIP(launch_gram(d_values + static_cast<size_t>(q0) * P * ix->dim, P, ix->dim, d_gram, m, s));
        RLR_HIP(launch_mmr_greedy(d_gram, d_sc, P, k, lambda, d_order, d_mmr, d_n, d_sizes, m, s));
        if (timed) RLR_HIP(hipEventRecord(c->ev[1], s));
        // results: order | mmr | n are contiguous
        uint32_t *h_res = reinterpret_cast<uint32_t *>(h_sizes + m + 4);
        RLR_HIP(hipMemcpyAsync(h_res, d_order, (2ull * n_list + m) * 4, hipMemcpyDeviceToHost, s));
        RLR_HIP(hipStreamSynchronize(s));
        if (timed)
            note_mmr(ix, c, m);
        const float *h_mmr = reinterpret_cast<const float *>(h_res + n_list);
        const uint32_t *h_n = h_res + 2ull * n_list;
        for (uint32_t q = 0; q < m; ++q) {
            const uint32_t ns = h_n[q];
            n_out[q0 + q] = ns;
            std::memcpy(order_out + static_cast<size_t>(q0 + q) * P, h_res + static_cast<size_t>(q) * P, ns * sizeof(uint32_t));
            if (mmr_out)
                std::memcpy(mmr_out + static_cast<size_t>(q0 + q) * P, h_mmr + static_cast<size_t>(q) * P, ns * sizeof(float));
        }
    }
    return RLR_OK;
}

int32_t rlr_mmr_select_batch(rlr_index *ix, const uint64_t *pool_rows, const float *pool_scores,
                             const uint32_t *pool_sizes, uint32_t n_queries, uint32_t P, uint32_t k, float lambda,
                             uint32_t *order_out, float *mmr_out, uint32_t *n_out)
{
    if (n_queries && !pool_rows)
        return fail(RLR_E_INVALID, "null argument");
    return mmr_batch_impl(ix, pool_rows, nullptr, pool_scores, pool_sizes, n_queries, P, k, lambda, order_out, mmr_out, n_out);
}

int32_t rlr_mmr_select_values(rlr_index *ix, const void *d_values, const float *pool_scores, const uint32_t *pool_sizes,
                              uint32_t n_queries, uint32_t P, uint32_t k, float lambda, uint32_t *order_out,
                              float *mmr_out, uint32_t *n_out)
{
    if (n_queries && !d_values)
        return fail(RLR_E_INVALID, "null argument");
    return mmr_batch_impl(ix, nullptr, static_cast<const float *>(d_values), pool_scores, pool_sizes, n_queries, P, k, lambda,
                          order_out, mmr_out, n_out);
}

int32_t rlr_mmr_select_staged(rlr_index *ix, const void *d_staged, uint64_t n_staged, const uint64_t *pool_slots,
                              const float *pool_scores, const uint32_t *pool_sizes, uint32_t n_queries, uint32_t P, uint32_t k,
                              float lambda, uint32_t *order_out, float *mmr_out, uint32_t *n_out)
{
    if (n_queries && (!d_staged || !pool_slots || n_staged == 0))
        return fail(RLR_E_INVALID, "null argument");
    if (n_queries == 1 && pool_sizes && pool_sizes[0] > 1024) // one large pool: the single-pool kernels (up to 4096)
        return mmr_single_impl(ix, pool_slots, pool_scores, pool_sizes[0], k, lambda, order_out, mmr_out, n_out, d_staged,
                               n_staged);
    return mmr_batch_impl(ix, pool_slots, nullptr, pool_scores, pool_sizes, n_queries, P, k, lambda, order_out, mmr_out, n_out,
                          d_staged, n_staged);
}

int32_t rlr_index_row_bytes(const rlr_index *ix, uint32_t *bytes_out)
{
    RLR_TRY(check_handle(ix));
    if (!bytes_out)
        return fail(RLR_E_INVALID, "bytes_out is null");
    *bytes_out = static_cast<uint32_t>(row_bytes(ix));
    return RLR_OK;
}

int32_t rlr_gather_rows_device(rlr_index *ix, const uint64_t *rows, uint32_t n, void *d_out)
{
    RLR_TRY(check_handle(ix));
    if (n == 0)
        return RLR_OK;
    if (!rows || !d_out)
        return fail(RLR_E_INVALID, "null argument");
    RLR_TRY(use_device(ix));
    CtxLease lease(ix);
    RLR_TRY(ctx_acquire(ix, &lease.c));
    Ctx *c = lease.c;
    RLR_TRY(upload_list(ix, c, rows, n));
    RLR_HIP(launch_compact_rows(ix->d_rows, d_out, ix->pitch16, c->d_list, n, c->stream));
    RLR_HIP(hipStreamSynchronize(c->stream));
    return RLR_OK;
}

int32_t rlr_fetch_rows_device(rlr_index *ix, const uint64_t *rows, uint32_t n, void *d_out)
{
    RLR_TRY(check_handle(ix));
    if (n == 0)
        return RLR_OK;
    if (!rows || !d_out)
        return fail(RLR_E_INVALID, "null argument");
    RLR_TRY(use_device(ix));
    CtxLease lease(ix);
    RLR_TRY(ctx_acquire(ix, &lease.c));
    Ctx *c = lease.c;
    RLR_TRY(upload_list(ix, c, rows, n));
    RLR_HIP(launch_gather_f32(ix->d_rows, ix->pitch16, ix->dim, ix->dtype, c->d_list, n, static_cast<float *>(d_out), c->stream));
    RLR_HIP(hipStreamSynchronize(c->stream));
    return RLR_OK;
}

int32_t rlr_index_probe_bandwidth(rlr_index *ix, int32_t mode, uint32_t reps, double *gbps_out, double *ms_out)
{
    RLR_TRY(check_handle(ix));
    if (!gbps_out || mode < 0 || mode > 3)
        return fail(RLR_E_INVALID, "mode must be 0 (read), 1 (copy), 2 / 3 (the scan kernel without / with its histogram), gbps_out non-null");
    *gbps_out = 0.0;
    if (ms_out)
        *ms_out = 0.0;
    RLR_TRY(use_device(ix));
    const size_t bytes = static_cast<size_t>(ix->n_rows) * row_bytes(ix);
    if (bytes < (1u << 20))
        return fail(RLR_E_INVALID, "the probe needs at least 1 MiB of rows");
    reps = std::max(reps, 1u);
    CtxLease lease(ix);
    RLR_TRY(ctx_acquire(ix, &lease.c));
    Ctx *c = lease.c;
    hipStream_t s = c->stream;
    void *scratch = nullptr;
    int32_t st = RLR_OK;
    double best_ms = 0.0;
    size_t moved = 0;
    auto timed = [&](auto &&launch) -> int32_t { // one warm-up, then `reps` launches between two events
        RLR_HIP(launch());
        RLR_HIP(hipEventRecord(c->ev[0], s));
        for (uint32_t i = 0; i < reps; ++i)
            RLR_HIP(launch());
        RLR_HIP(hipEventRecord(c->ev[1], s));
        RLR_HIP(hipStreamSynchronize(s));
        float ms = 0;
        RLR_HIP(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        const double per = static_cast<double>(ms) / reps;
        if (best_ms == 0.0 || per < best_ms)
            best_ms = per;
        return RLR_OK;
    };
    if (mode == 0) {
        RLR_HIP(rlr::dev_malloc(&scratch, static_cast<size_t>(ix->n_cu) * 8 * 256 * sizeof(float)));
        // (experiments: RLR_PROBE_SHAPE pins the launch shape, RLR_PROBE_OFF_MIB / RLR_PROBE_LEN_MIB a sub-range of the rows)
        const char *es = getenv("RLR_PROBE_SHAPE"), *eo = getenv("RLR_PROBE_OFF_MIB"), *el = getenv("RLR_PROBE_LEN_MIB");
        size_t off = eo ? static_cast<size_t>(strtoull(eo, nullptr, 10)) << 20 : 0;
        size_t len = el ? static_cast<size_t>(strtoull(el, nullptr, 10)) << 20 : bytes;
        off = std::min(off, bytes - (1u << 20));
        len = std::min(len, bytes - off);
        moved = len / 1024 * 1024;
        const char *base = static_cast<const char *>(ix->d_rows) + off;
        for (int shape = es ? atoi(es) : 0; shape < (es ? atoi(es) + 1 : 3) && st == RLR_OK; ++shape)
            st = timed([&] { return launch_probe_read(base, len, static_cast<float *>(scratch), ix->n_cu, shape, s); });
    } else if (mode >= 2) {
        // diagnostic: the scan kernel itself over the rows with a zero query, scores into a scratch array, without (2) or
        // with (3) the digit-1 histogram it accumulates in LDS and flushes with global atomics
        const size_t sc_bytes = (static_cast<size_t>(ix->n_rows) + 2 * kHistBins + ix->q_pitch) * sizeof(float);
        RLR_HIP(rlr::dev_malloc(&scratch, sc_bytes));
        RLR_HIP(hipMemsetAsync(scratch, 0, sc_bytes, s));
        const char *eo = getenv("RLR_PROBE_OFF_MIB"), *el = getenv("RLR_PROBE_LEN_MIB"); // (experiments: a sub-range of the rows)
        const uint64_t row_lo = eo ? std::min<uint64_t>((strtoull(eo, nullptr, 10) << 20) / row_bytes(ix), ix->n_rows - 1) : 0;
        const uint64_t row_n = el ? std::min<uint64_t>((strtoull(el, nullptr, 10) << 20) / row_bytes(ix), ix->n_rows - row_lo)
                                  : ix->n_rows - row_lo;
        ScanArgs sa;
        sa.rows = static_cast<const char *>(ix->d_rows) + row_lo * row_bytes(ix);
        sa.scores = static_cast<float *>(scratch);
        sa.hist = mode == 3 ? reinterpret_cast<uint32_t *>(sa.scores + ix->n_rows) : nullptr;
        sa.query = sa.scores + ix->n_rows + 2 * kHistBins;
        sa.n_rows = static_cast<uint32_t>(row_n);
        sa.dim = ix->dim;
        sa.pitch16 = ix->pitch16;
        sa.dtype = ix->dtype;
        sa.n_cu = ix->n_cu;
        sa.variant = ix->scan_variant;
        moved = static_cast<size_t>(row_n) * ix->dim * (ix->dtype == RLR_F16 ? 2 : 4);
        st = timed([&] { return launch_scan(sa, s); });
    } else {
        const size_t half = std::min<size_t>(bytes / 2, 4ull << 30) & ~static_cast<size_t>(255);
        hipError_t e = rlr::dev_malloc(&scratch, half);
        if (e != hipSuccess)
            return fail(RLR_E_OOM, "scratch allocation of %zu bytes for the copy probe failed", half);
        moved = 2 * half;
        st = timed([&] { return hipMemcpyAsync(scratch, ix->d_rows, half, hipMemcpyDeviceToDevice, s); });
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(scratch);
    if (st != RLR_OK)
        return st;
    *gbps_out = static_cast<double>(moved) / (best_ms * 1e-3) / 1e9;
    if (ms_out)
        *ms_out = best_ms;
    return RLR_OK;
}

int32_t rlr_profile_enable(rlr_index *ix, int32_t enable)
{
    RLR_TRY(check_handle(ix));
    ix->profiling = enable != 0;
    return RLR_OK;
}

int32_t rlr_profile_read(rlr_index *ix, rlr_profile *out, int32_t reset)
{
    RLR_TRY(check_handle(ix));
    if (!out)
        return fail(RLR_E_INVALID, "out is null");
    std::lock_guard<std::mutex> lk(ix->mu);
    *out = ix->prof;
    if (reset)
        ix->prof = rlr_profile{};
    return RLR_OK;
}

} // extern "C"
