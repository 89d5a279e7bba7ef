// multi.cpp -- one process driving several GPUs: contiguous row shards, one PERSISTENT host thread per shard
// (created with the handle, parked on a condition variable between calls), and two forms of the exchange step
// of SURVEY.md 8(e):
//   * host merge (default): every shard's k x (row, score) list comes back to the host, k-way merge there;
//   * RCCL (rlr_multi_set_exchange(m, 1)): every shard leaves its packed partial top-k in device memory
//     (rlr_search_topk_device), one ncclAllGather of n_queries x k x 8 bytes per shard over xGMI inside a
//     group call, merge_topk_kernel on the first device (rlr_merge_topk) -- the path BASELINE.json's north_star
//     names, reachable from the C ABI without torch or one-process-per-GPU.  librccl is loaded with dlopen on
//     first use, so the library has no link-time dependency on it.
// Cross-shard MMR (SURVEY.md 8(e) "MMR on sharded data"): the pool rows of a query live on several GPUs; every shard
// gathers the rows it owns on its own device (rlr_gather_rows_device: raw rows, binary16 stays binary16), copies them
// device to device (hipMemcpyPeerAsync: xGMI, no host bounce) to the GPU that owns the query -- queries are dealt
// round-robin over the shards -- and that GPU runs the Gram + greedy kernels over its receive buffer
// (rlr_mmr_select_staged).  The engine-level entry points (rlr_multi_engine_*: RagEngine::search /
// search_with_diversity over the sharded corpus) are csrc/engine_host.h instantiated on these primitives.
// Built on the single-index C ABI plus HIP runtime calls for the exchange buffers; no device code here.
#include "../../include/rlr_gpu.h"
#include "../../include/rlr_engine.h"
#include "../../include/rlr_lexical.h"
#include "engine_host.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace rlr {
int32_t set_error(int32_t code, const char *fmt, ...); // index.hip: the calling thread's rlr_last_error() text
hipError_t dev_malloc(void **p, size_t bytes);         // index.hip: every device allocation (RLR_POISON_ALLOC covers it)
}

namespace {

// One parked thread per shard: jobs are closures, a call waits on its own latch, so concurrent callers
// (the reference's concurrent readers) interleave on the workers instead of excluding each other.
struct Latch {
    std::mutex mu;
    std::condition_variable cv;
    size_t remaining = 0;
};

struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> jobs;
    bool quit = false;

    void start()
    {
        th = std::thread([this] {
            for (;;) {
                std::function<void()> job;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [this] { return quit || !jobs.empty(); });
                    if (jobs.empty())
                        return; // quit
                    job = std::move(jobs.front());
                    jobs.pop_front();
                }
                job();
            }
        });
    }
    void post(std::function<void()> job)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            jobs.push_back(std::move(job));
        }
        cv.notify_one();
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv.notify_one();
        if (th.joinable())
            th.join();
    }
};

// librccl through dlopen: only the five entry points the exchange needs
struct Rccl {
    void *so = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.so)
                break;
        }
        if (!r.so)
            return;
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.so, "ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.so, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.so, "ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.so, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.so, "ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.so, "ncclGetErrorString"));
        r.ok = r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd && r.GetErrorString;
    });
    return r;
}

} // namespace

struct rlr_multi {
    uint32_t dim = 0;
    int32_t dtype = RLR_F32;
    std::vector<rlr_index *> shard;
    std::vector<int32_t> device;
    std::vector<uint64_t> base; // first global row of each shard (+ total at the end)
    uint64_t n_rows = 0;
    std::vector<Worker *> worker; // shards 1..G-1 (shard 0 runs on the calling thread)
    // RCCL exchange (rlr_multi_set_exchange)
    int32_t exchange = 0;
    std::mutex xmu; // one collective at a time is ENQUEUED on the communicators (the scans around it are not under it)
    std::vector<ncclComm_t> comm;
    // exchange workspaces (a stream, the local list and the gathered lists per shard), leased per call like XferWs:
    // concurrent callers scan side by side and only take turns for the group call itself
    std::mutex emu;
    std::condition_variable ecv;
    std::vector<struct ExWs *> e_free;
    int e_made = 0;
    // cross-shard MMR: a few exchange workspaces (stage + receive buffer and a copy stream per shard), leased per call
    std::mutex wmu;
    std::condition_variable wcv;
    std::vector<struct XferWs *> w_free;
    int w_made = 0;
    uint32_t row_bytes = 0;
    // which path the calls took (rlr_multi_stats)
    std::atomic<uint64_t> n_topk_rccl{0}, n_topk_host{0}, n_topk_fellback{0}, n_mmr_exchanges{0}, mmr_exchange_bytes{0};
    std::atomic<uint64_t> topk_exchange_ns{0}, mmr_exchange_ns{0};
    std::atomic<uint64_t> n_mmr_host_bounces{0}; // winner-row pieces that went through the host because the peer copy failed
};

struct ExWs {
    std::vector<hipStream_t> stream;
    std::vector<void *> d_local, d_gath;
    size_t cap = 0; // entries (u64) d_local holds per shard
};

constexpr int kMaxExWs = 4;

struct XferWs {
    std::vector<void *> d_stage, d_recv; // per shard, on its device
    std::vector<size_t> stage_cap, recv_cap;
    std::vector<hipStream_t> stream;
};

constexpr int kMaxXferWs = 4; // concurrent cross-shard MMR calls beyond this wait

namespace {

void shard_of(const rlr_multi *m, uint64_t row, uint32_t *s, uint64_t *local)
{
    uint32_t g = static_cast<uint32_t>(std::upper_bound(m->base.begin(), m->base.end(), row) - m->base.begin()) - 1;
    if (g >= m->shard.size())
        g = static_cast<uint32_t>(m->shard.size()) - 1;
    *s = g;
    *local = row - m->base[g];
}

void set_bases(rlr_multi *m, uint64_t n_rows)
{
    const uint64_t G = m->shard.size();
    const uint64_t per = (n_rows + G - 1) / G;
    m->base.assign(G + 1, 0);
    for (uint64_t g = 0; g <= G; ++g)
        m->base[g] = std::min<uint64_t>(n_rows, g * per);
    m->n_rows = n_rows;
}

// run f(g) for every shard on its persistent thread (shard 0 on the caller's); returns the first failing status.
// rlr_last_error() is per thread, so a worker's message is carried back to the calling thread (prefixed with the
// shard it came from).
template <typename F>
int32_t for_each_shard(const rlr_multi *m, F f)
{
    const size_t G = m->shard.size();
    std::vector<int32_t> st(G, RLR_OK);
    if (G == 1) {
        st[0] = f(0);
        return st[0];
    }
    std::vector<std::string> why(G);
    Latch latch;
    latch.remaining = G - 1;
    for (size_t g = 1; g < G; ++g)
        m->worker[g - 1]->post([&, g] {
            st[g] = f(static_cast<uint32_t>(g));
            if (st[g] != RLR_OK)
                why[g] = rlr_last_error();
            std::lock_guard<std::mutex> lk(latch.mu);
            if (--latch.remaining == 0)
                latch.cv.notify_one();
        });
    st[0] = f(0);
    if (st[0] != RLR_OK)
        why[0] = rlr_last_error();
    {
        std::unique_lock<std::mutex> lk(latch.mu);
        latch.cv.wait(lk, [&] { return latch.remaining == 0; });
    }
    for (size_t g = 0; g < G; ++g)
        if (st[g] != RLR_OK)
            return rlr::set_error(st[g], "shard %zu: %s", g, why[g].c_str());
    return RLR_OK;
}

void exws_destroy(rlr_multi *m, ExWs *w)
{
    if (!w)
        return;
    for (size_t g = 0; g < w->stream.size(); ++g) {
        (void)hipSetDevice(m->device[g]);
        if (w->stream[g]) {
            (void)hipStreamSynchronize(w->stream[g]);
            (void)hipStreamDestroy(w->stream[g]);
        }
        if (w->d_local[g]) (void)hipFree(w->d_local[g]);
        if (w->d_gath[g]) (void)hipFree(w->d_gath[g]);
    }
    delete w;
}

void exchange_teardown(rlr_multi *m)
{
    for (ExWs *w : m->e_free)
        exws_destroy(m, w);
    m->e_free.clear();
    m->e_made = 0;
    for (size_t g = 0; g < m->comm.size(); ++g)
        if (m->comm[g])
            (void)rccl().CommDestroy(m->comm[g]);
    m->comm.clear();
}

#define RLR_X_HIP(call)                                                                                        \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return rlr::set_error(RLR_E_HIP, "%s: %s", #call, hipGetErrorString(e_));                          \
    } while (0)
#define RLR_X_NCCL(call)                                                                                       \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess)                                                                                 \
            return rlr::set_error(RLR_E_HIP, "%s: %s", #call, rccl().GetErrorString(r_));                      \
    } while (0)

int32_t exws_acquire(rlr_multi *m, ExWs **out)
{
    {
        std::unique_lock<std::mutex> lk(m->emu);
        for (;;) {
            if (!m->e_free.empty()) {
                *out = m->e_free.back();
                m->e_free.pop_back();
                return RLR_OK;
            }
            if (m->e_made < kMaxExWs) {
                m->e_made++;
                break;
            }
            m->ecv.wait(lk);
        }
    }
    const size_t G = m->shard.size();
    ExWs *w = new (std::nothrow) ExWs();
    hipError_t e = w ? hipSuccess : hipErrorOutOfMemory;
    if (w) {
        w->stream.assign(G, nullptr);
        w->d_local.assign(G, nullptr);
        w->d_gath.assign(G, nullptr);
        for (size_t g = 0; g < G && e == hipSuccess; ++g) {
            e = hipSetDevice(m->device[g]);
            if (e == hipSuccess)
                e = hipStreamCreateWithFlags(&w->stream[g], hipStreamNonBlocking);
        }
    }
    if (e != hipSuccess) {
        exws_destroy(m, w);
        {
            std::lock_guard<std::mutex> lk(m->emu);
            m->e_made--;
        }
        m->ecv.notify_one();
        return rlr::set_error(RLR_E_HIP, "exchange workspace: %s", hipGetErrorString(e));
    }
    *out = w;
    return RLR_OK;
}

void exws_release(rlr_multi *m, ExWs *w)
{
    {
        std::lock_guard<std::mutex> lk(m->emu);
        m->e_free.push_back(w);
    }
    m->ecv.notify_one();
}

struct ExLease {
    rlr_multi *m;
    ExWs *w = nullptr;
    explicit ExLease(rlr_multi *mm) : m(mm) {}
    ~ExLease()
    {
        if (w)
            exws_release(m, w);
    }
};

// The RCCL form of the exchange step.  *handled = false: a shard's guard band overflowed (or the shape is outside
// the merge kernel) and the caller falls through to the host merge, which handles everything.
//
// No host round trip between the scans and the collective: every shard ENQUEUES its pipelines on the workspace's stream
// of its device (rlr_search_topk_device_begin), the all-gathers queue up behind them in one group call, the merge
// kernel behind device 0's; the one wait of the call is the merge's.  The workspace (streams + lists) is leased per
// call, so concurrent callers overlap their scans and hold `xmu` only while the group call is enqueued -- a
// communicator takes one collective at a time.  Nothing waits for the other devices' halves of the collective: the
// next user of the workspace enqueues behind them on the same streams.
int32_t search_rccl(rlr_multi *m, const float *queries, uint32_t nq, uint32_t k, float guard_eps, uint64_t *rows_out,
                    float *cos_out, uint32_t *n_out, bool *handled)
{
    *handled = false;
    const size_t G = m->shard.size();
    if (G > 16 || static_cast<uint64_t>(G) * k > 8192 || nq == 0 || k == 0)
        return RLR_OK;
    ExLease lease(m);
    {
        const int32_t as = exws_acquire(m, &lease.w);
        if (as != RLR_OK)
            return as;
    }
    ExWs *w = lease.w;
    const size_t per = static_cast<size_t>(nq) * k;
    if (w->cap < per) {
        w->cap = 0; // nothing usable until every buffer of the new size exists
        for (size_t g = 0; g < G; ++g) {
            RLR_X_HIP(hipSetDevice(m->device[g]));
            RLR_X_HIP(hipStreamSynchronize(w->stream[g])); // (an earlier call's collective may still read the old buffers)
            if (w->d_local[g]) (void)hipFree(w->d_local[g]);
            if (w->d_gath[g]) (void)hipFree(w->d_gath[g]);
            w->d_local[g] = w->d_gath[g] = nullptr;
            RLR_X_HIP(rlr::dev_malloc(&w->d_local[g], per * sizeof(uint64_t)));
            RLR_X_HIP(rlr::dev_malloc(&w->d_gath[g], G * per * sizeof(uint64_t)));
        }
        w->cap = per;
    }
    // every shard: the whole local pipeline enqueued, k packed results per query into its device memory
    std::vector<void *> ticket(G, nullptr);
    auto end_all = [&]() { // hand the search contexts back (waits for whatever is still in flight on the shard's stream)
        uint32_t over = 0;
        int32_t est = RLR_OK;
        for (size_t g = 0; g < G; ++g) {
            uint32_t o = 0;
            const int32_t s1 = rlr_search_topk_device_end(m->shard[g], ticket[g], &o);
            ticket[g] = nullptr;
            over += o;
            if (s1 != RLR_OK && est == RLR_OK)
                est = s1;
        }
        return std::make_pair(est, over);
    };
    int32_t st = for_each_shard(m, [&](uint32_t g) {
        return rlr_search_topk_device_begin(m->shard[g], queries, nq, k, guard_eps, w->d_local[g], w->stream[g], &ticket[g]);
    });
    if (st != RLR_OK) {
        (void)end_all();
        return st;
    }
    // one all-gather per shard inside a group call: G x nq x k x 8 bytes land on every device, rank-major
    {
        std::lock_guard<std::mutex> xl(m->xmu);
        ncclResult_t r = rccl().GroupStart();
        for (size_t g = 0; g < G && r == ncclSuccess; ++g)
            r = rccl().AllGather(w->d_local[g], w->d_gath[g], per, ncclUint64, m->comm[g], w->stream[g]);
        const ncclResult_t r2 = rccl().GroupEnd();
        if (r == ncclSuccess)
            r = r2;
        if (r != ncclSuccess) {
            (void)end_all();
            return rlr::set_error(RLR_E_HIP, "ncclAllGather group: %s", rccl().GetErrorString(r));
        }
    }
    // merge on the first device (a single process needs the answer once); queued behind its all-gather
    st = rlr_merge_topk(m->device[0], w->d_gath[0], static_cast<uint32_t>(G), nq, k, m->base.data(), rows_out, cos_out,
                        n_out, w->stream[0]);
    const auto ended = end_all();
    if (st != RLR_OK)
        return st;
    if (ended.first != RLR_OK)
        return ended.first;
    for (uint32_t q = 0; q < nq; ++q)
        if (n_out[q] == 0xFFFFFFFFu)
            return RLR_OK; // overflow marker: host merge redoes the call
    *handled = true;
    return RLR_OK;
}


void xfer_destroy(rlr_multi *m, XferWs *w)
{
    if (!w)
        return;
    for (size_t g = 0; g < w->stream.size(); ++g) {
        (void)hipSetDevice(m->device[g]);
        if (w->stream[g]) {
            (void)hipStreamSynchronize(w->stream[g]);
            (void)hipStreamDestroy(w->stream[g]);
        }
        if (w->d_stage[g]) (void)hipFree(w->d_stage[g]);
        if (w->d_recv[g]) (void)hipFree(w->d_recv[g]);
    }
    delete w;
}

int32_t xfer_acquire(rlr_multi *m, XferWs **out)
{
    std::unique_lock<std::mutex> lk(m->wmu);
    for (;;) {
        if (!m->w_free.empty()) {
            *out = m->w_free.back();
            m->w_free.pop_back();
            return RLR_OK;
        }
        if (m->w_made < kMaxXferWs) {
            m->w_made++;
            break;
        }
        m->wcv.wait(lk);
    }
    lk.unlock();
    const size_t G = m->shard.size();
    XferWs *w = new XferWs();
    w->d_stage.assign(G, nullptr);
    w->d_recv.assign(G, nullptr);
    w->stage_cap.assign(G, 0);
    w->recv_cap.assign(G, 0);
    w->stream.assign(G, nullptr);
    for (size_t g = 0; g < G; ++g) {
        hipError_t e = hipSetDevice(m->device[g]);
        if (e == hipSuccess)
            e = hipStreamCreateWithFlags(&w->stream[g], hipStreamNonBlocking);
        if (e != hipSuccess) {
            xfer_destroy(m, w);
            {
                std::lock_guard<std::mutex> lk2(m->wmu);
                m->w_made--;
            }
            m->wcv.notify_one();
            return rlr::set_error(RLR_E_HIP, "exchange stream on device %d: %s", m->device[g], hipGetErrorString(e));
        }
    }
    *out = w;
    return RLR_OK;
}

void xfer_release(rlr_multi *m, XferWs *w)
{
    {
        std::lock_guard<std::mutex> lk(m->wmu);
        m->w_free.push_back(w);
    }
    m->wcv.notify_one();
}

struct XferLease {
    rlr_multi *m;
    XferWs *w = nullptr;
    ~XferLease()
    {
        if (w)
            xfer_release(m, w);
    }
};

int32_t dev_reserve(int32_t device, void **p, size_t *cap, size_t bytes)
{
    if (*cap >= bytes && *p)
        return RLR_OK;
    RLR_X_HIP(hipSetDevice(device));
    if (*p)
        (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 4, 1 << 20);
    RLR_X_HIP(rlr::dev_malloc(p, want));
    *cap = want;
    return RLR_OK;
}

// mmr_diversify over nq pools (strided by P) whose rows are GLOBAL rows of the sharded corpus.  Query q is diversified
// on shard q mod G; each shard gathers the pool rows it owns, device-to-device copies put them into the owners' receive
// buffers (grouped by source shard), and every owner runs the batched Gram + greedy kernels over its buffer through a
// slot list that restores the pool order.  Same results as rlr_mmr_select_batch over one index holding all rows.
int32_t multi_mmr(rlr_multi *m, const uint64_t *pool_rows, const float *pool_scores, const uint32_t *pool_sizes, uint32_t nq,
                  uint32_t P, uint32_t k, float lambda, uint32_t *order_out, float *mmr_out, uint32_t *n_sel)
{
    const size_t G = m->shard.size();
    for (uint32_t q = 0; q < nq; ++q)
        n_sel[q] = 0;
    if (nq == 0 || P == 0)
        return RLR_OK;
    if (G == 1) {
        if (nq == 1)
            return rlr_mmr_select(m->shard[0], pool_rows, pool_scores, pool_sizes[0], k, lambda, order_out, mmr_out, n_sel);
        return rlr_mmr_select_batch(m->shard[0], pool_rows, pool_scores, pool_sizes, nq, P, k, lambda, order_out, mmr_out, n_sel);
    }
    if (nq > 1 && P > 1024)
        return rlr::set_error(RLR_E_INVALID, "batched MMR supports pools of at most 1024 candidates (got %u)", P);
    const auto t0 = std::chrono::steady_clock::now();
    const size_t rb = m->row_bytes;
    // plan: entries of shard g ordered by (owner, query, position); owner o receives the blocks in source-shard order
    std::vector<std::vector<uint64_t>> send_rows(G);              // local rows, in send order
    std::vector<std::vector<size_t>> send_count(G, std::vector<size_t>(G, 0)); // [g][o]
    for (uint32_t q = 0; q < nq; ++q) {
        if (pool_sizes[q] > P)
            return rlr::set_error(RLR_E_INVALID, "pool_sizes[%u] = %u exceeds P = %u", q, pool_sizes[q], P);
        for (uint32_t j = 0; j < pool_sizes[q]; ++j) {
            const uint64_t row = pool_rows[static_cast<size_t>(q) * P + j];
            if (row >= m->n_rows)
                return rlr::set_error(RLR_E_RANGE, "row %llu out of range", static_cast<unsigned long long>(row));
            uint32_t g;
            uint64_t l;
            shard_of(m, row, &g, &l);
            send_count[g][q % G]++;
        }
    }
    std::vector<std::vector<size_t>> send_off(G, std::vector<size_t>(G + 1, 0)), recv_off(G, std::vector<size_t>(G + 1, 0));
    for (size_t g = 0; g < G; ++g)
        for (size_t o = 0; o < G; ++o)
            send_off[g][o + 1] = send_off[g][o] + send_count[g][o];
    for (size_t o = 0; o < G; ++o)
        for (size_t g = 0; g < G; ++g)
            recv_off[o][g + 1] = recv_off[o][g] + send_count[g][o];
    for (size_t g = 0; g < G; ++g)
        send_rows[g].assign(send_off[g][G], 0);
    // owners' compact query lists and slot tables
    std::vector<std::vector<uint32_t>> owned(G);
    for (uint32_t q = 0; q < nq; ++q)
        owned[q % G].push_back(q);
    std::vector<std::vector<uint64_t>> slots(G);
    std::vector<std::vector<float>> scores(G);
    std::vector<std::vector<uint32_t>> sizes(G);
    for (size_t o = 0; o < G; ++o) {
        slots[o].assign(owned[o].size() * static_cast<size_t>(P), 0);
        scores[o].assign(owned[o].size() * static_cast<size_t>(P), 0.0f);
        sizes[o].resize(owned[o].size());
    }
    std::vector<std::vector<size_t>> fill(G, std::vector<size_t>(G, 0)); // [g][o]: entries placed so far
    for (size_t o = 0; o < G; ++o)
        for (size_t i = 0; i < owned[o].size(); ++i) {
            const uint32_t q = owned[o][i];
            sizes[o][i] = pool_sizes[q];
            for (uint32_t j = 0; j < pool_sizes[q]; ++j) {
                uint32_t g;
                uint64_t l;
                shard_of(m, pool_rows[static_cast<size_t>(q) * P + j], &g, &l);
                const size_t at = fill[g][o]++;
                send_rows[g][send_off[g][o] + at] = l;
                slots[o][i * P + j] = recv_off[o][g] + at;
                scores[o][i * P + j] = pool_scores[static_cast<size_t>(q) * P + j];
            }
        }
    XferLease lease{m};
    int32_t st = xfer_acquire(m, &lease.w);
    if (st != RLR_OK)
        return st;
    XferWs *w = lease.w;
    uint64_t moved = 0;
    for (size_t g = 0; g < G; ++g) { // buffers first: a copy must never target memory that is still being (re)allocated
        if ((st = dev_reserve(m->device[g], &w->d_stage[g], &w->stage_cap[g], send_off[g][G] * rb)) != RLR_OK)
            return st;
        if ((st = dev_reserve(m->device[g], &w->d_recv[g], &w->recv_cap[g], recv_off[g][G] * rb)) != RLR_OK)
            return st;
        moved += send_off[g][G] * rb;
    }
    // 1. gather on the owning device, then one device-to-device copy per (source, owner) pair
    st = for_each_shard(m, [&](uint32_t g) -> int32_t {
        const size_t n_g = send_off[g][G];
        if (n_g == 0)
            return RLR_OK;
        const int32_t rc = [&]() -> int32_t {
            int32_t s1 = rlr_gather_rows_device(m->shard[g], send_rows[g].data(), static_cast<uint32_t>(n_g), w->d_stage[g]);
            if (s1 != RLR_OK)
                return s1;
            RLR_X_HIP(hipSetDevice(m->device[g]));
            for (size_t o = 0; o < G; ++o) {
                const size_t cnt = send_count[g][o];
                if (cnt == 0)
                    continue;
                char *dst = static_cast<char *>(w->d_recv[o]) + recv_off[o][g] * rb;
                const char *src = static_cast<const char *>(w->d_stage[g]) + send_off[g][o] * rb;
                if (m->device[o] == m->device[g]) {
                    RLR_X_HIP(hipMemcpyAsync(dst, src, cnt * rb, hipMemcpyDeviceToDevice, w->stream[g]));
                } else if (hipMemcpyPeerAsync(dst, m->device[o], src, m->device[g], cnt * rb, w->stream[g]) != hipSuccess) {
                    // no peer path between this pair (or the runtime refused it): bounce the piece through the host
                    // rather than failing the whole diversified search -- slower, same bytes
                    (void)hipGetLastError();
                    std::vector<char> bounce(cnt * rb);
                    RLR_X_HIP(hipStreamSynchronize(w->stream[g])); // (the gather that produced `src`)
                    RLR_X_HIP(hipMemcpy(bounce.data(), src, cnt * rb, hipMemcpyDeviceToHost));
                    RLR_X_HIP(hipSetDevice(m->device[o]));
                    RLR_X_HIP(hipMemcpy(dst, bounce.data(), cnt * rb, hipMemcpyHostToDevice));
                    RLR_X_HIP(hipSetDevice(m->device[g]));
                    m->n_mmr_host_bounces++;
                }
            }
            return RLR_OK;
        }();
        // copies already queued must have landed before the workspace can go back to the pool, whatever happened after them
        const hipError_t e = hipStreamSynchronize(w->stream[g]);
        if (rc != RLR_OK)
            return rc;
        if (e != hipSuccess)
            return rlr::set_error(RLR_E_HIP, "winner-row exchange on device %d: %s", m->device[g], hipGetErrorString(e));
        return RLR_OK;
    });
    if (st != RLR_OK)
        return st;
    m->n_mmr_exchanges++;
    m->mmr_exchange_bytes += moved;
    m->mmr_exchange_ns += static_cast<uint64_t>(
        std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
    // 2. every owner: Gram + greedy over its receive buffer
    std::vector<std::vector<uint32_t>> ord(G), nsel(G);
    std::vector<std::vector<float>> mmrv(G);
    st = for_each_shard(m, [&](uint32_t o) -> int32_t {
        const uint32_t nqo = static_cast<uint32_t>(owned[o].size());
        if (nqo == 0)
            return RLR_OK;
        ord[o].assign(static_cast<size_t>(nqo) * P, 0);
        mmrv[o].assign(static_cast<size_t>(nqo) * P, 0.0f);
        nsel[o].assign(nqo, 0);
        if (recv_off[o][G] == 0)
            return RLR_OK; // (only empty pools)
        return rlr_mmr_select_staged(m->shard[o], w->d_recv[o], recv_off[o][G], slots[o].data(), scores[o].data(), sizes[o].data(),
                                     nqo, P, k, lambda, ord[o].data(), mmrv[o].data(), nsel[o].data());
    });
    if (st != RLR_OK)
        return st;
    for (size_t o = 0; o < G; ++o)
        for (size_t i = 0; i < owned[o].size(); ++i) {
            const uint32_t q = owned[o][i];
            n_sel[q] = nsel[o][i];
            std::memcpy(order_out + static_cast<size_t>(q) * P, ord[o].data() + i * P, nsel[o][i] * sizeof(uint32_t));
            if (mmr_out)
                std::memcpy(mmr_out + static_cast<size_t>(q) * P, mmrv[o].data() + i * P, nsel[o][i] * sizeof(float));
        }
    return RLR_OK;
}

// the sharded corpus as a backend of engine_host.h
struct MultiBackend {
    rlr_multi *m;
    uint64_t n_rows;
    uint32_t dim;
    int32_t topk(const float *queries, uint32_t nq, uint32_t k, uint64_t *rows, float *cos, uint32_t *n) const
    {
        return rlr_multi_search_topk(m, queries, nq, k, -1.0f, rows, cos, n);
    }
    int32_t score_rows(const float *query, const uint64_t *rows, uint32_t n, float *cos) const
    {
        return rlr_multi_score_rows(m, query, rows, n, cos);
    }
    int32_t mmr(const uint64_t *pool_rows, const float *pool_scores, const uint32_t *pool_sizes, uint32_t nq, uint32_t P,
                uint32_t k, float lambda, uint32_t *order, uint32_t *n_sel) const
    {
        return multi_mmr(m, pool_rows, pool_scores, pool_sizes, nq, P, k, lambda, order, nullptr, n_sel);
    }
};

// (score desc, NaN last, global row asc)
bool hit_before(float sa, uint64_t ra, float sb, uint64_t rb)
{
    const bool an = std::isnan(sa), bn = std::isnan(sb);
    if (an || bn) {
        if (an != bn)
            return bn;
        return ra < rb;
    }
    if (sa != sb)
        return sa > sb;
    return ra < rb;
}

} // namespace

extern "C" {

int32_t rlr_multi_create(uint32_t dim, int32_t dtype, int32_t n_devices, const int32_t *device_ids, rlr_multi **out)
{
    if (!out || n_devices <= 0 || !device_ids)
        return RLR_E_INVALID;
    *out = nullptr;
    rlr_multi *m = new rlr_multi();
    m->dim = dim;
    m->dtype = dtype;
    for (int32_t g = 0; g < n_devices; ++g) {
        rlr_index *ix = nullptr;
        const int32_t st = rlr_index_create(dim, dtype, device_ids[g], &ix);
        if (st != RLR_OK) {
            rlr_multi_destroy(m);
            return st;
        }
        m->shard.push_back(ix);
        m->device.push_back(device_ids[g]);
    }
    for (int32_t g = 1; g < n_devices; ++g) {
        Worker *w = new Worker();
        w->start();
        m->worker.push_back(w);
    }
    const int32_t st = rlr_index_row_bytes(m->shard[0], &m->row_bytes);
    if (st != RLR_OK) {
        rlr_multi_destroy(m);
        return st;
    }
    // the winner-row exchange copies device to device: map the peers where the platform allows it (a refusal only
    // means the runtime stages the copy itself)
    int caller_device = 0;
    const bool have_caller_device = hipGetDevice(&caller_device) == hipSuccess;
    for (int32_t a = 0; a < n_devices; ++a)
        for (int32_t b = 0; b < n_devices; ++b) {
            int can = 0;
            if (device_ids[a] == device_ids[b] || hipDeviceCanAccessPeer(&can, device_ids[a], device_ids[b]) != hipSuccess || !can)
                continue;
            if (hipSetDevice(device_ids[a]) == hipSuccess && hipDeviceEnablePeerAccess(device_ids[b], 0) != hipSuccess)
                (void)hipGetLastError(); // (already enabled counts as an error)
        }
    if (have_caller_device)
        (void)hipSetDevice(caller_device); // (the loop above moved the calling thread's current device)
    set_bases(m, 0);
    *out = m;
    return RLR_OK;
}

int32_t rlr_multi_destroy(rlr_multi *m)
{
    if (!m)
        return RLR_OK;
    for (Worker *w : m->worker) {
        w->stop();
        delete w;
    }
    exchange_teardown(m);
    for (XferWs *w : m->w_free)
        xfer_destroy(m, w);
    for (rlr_index *ix : m->shard)
        rlr_index_destroy(ix);
    delete m;
    return RLR_OK;
}

int32_t rlr_multi_set_exchange(rlr_multi *m, int32_t mode)
{
    if (!m || (mode != 0 && mode != 1))
        return RLR_E_INVALID;
    if (mode == 0) {
        std::lock_guard<std::mutex> xl(m->xmu);
        m->exchange = 0;
        return RLR_OK;
    }
    const size_t G = m->shard.size();
    for (size_t a = 0; a < G; ++a)
        for (size_t b = a + 1; b < G; ++b)
            if (m->device[a] == m->device[b])
                return rlr::set_error(RLR_E_INVALID, "RCCL exchange needs one shard per device (device %d holds two)", m->device[a]);
    if (!rccl().ok) {
        const char *why = dlerror(); // (reading it clears it)
        return rlr::set_error(RLR_E_NO_DEVICE, "librccl.so could not be loaded: %s", why ? why : "symbols missing");
    }
    std::lock_guard<std::mutex> xl(m->xmu);
    if (m->comm.empty()) {
        m->comm.assign(G, nullptr);
        const ncclResult_t r = rccl().CommInitAll(m->comm.data(), static_cast<int>(G), m->device.data());
        if (r != ncclSuccess) {
            m->comm.clear();
            return rlr::set_error(RLR_E_HIP, "ncclCommInitAll: %s", rccl().GetErrorString(r));
        }
    }
    m->exchange = 1;
    return RLR_OK;
}

int32_t rlr_multi_info(const rlr_multi *m, uint64_t *n_rows, uint32_t *n_shards)
{
    if (!m)
        return RLR_E_INVALID;
    if (n_rows) *n_rows = m->n_rows;
    if (n_shards) *n_shards = static_cast<uint32_t>(m->shard.size());
    return RLR_OK;
}

int32_t rlr_multi_upload(rlr_multi *m, const float *rows, uint64_t n_rows, int32_t normalize_on_device)
{
    if (!m || (n_rows && !rows))
        return RLR_E_INVALID;
    set_bases(m, n_rows);
    return for_each_shard(m, [&](uint32_t g) {
        const uint64_t lo = m->base[g], hi = m->base[g + 1];
        return rlr_index_upload(m->shard[g], rows + lo * m->dim, hi - lo, normalize_on_device);
    });
}

int32_t rlr_multi_fill_synthetic(rlr_multi *m, uint64_t n_rows, uint64_t seed, uint32_t n_clusters)
{
    if (!m)
        return RLR_E_INVALID;
    set_bases(m, n_rows);
    return for_each_shard(m, [&](uint32_t g) {
        const uint64_t lo = m->base[g], hi = m->base[g + 1];
        return rlr_index_fill_synthetic(m->shard[g], hi - lo, lo, seed, n_clusters);
    });
}

int32_t rlr_multi_enable_batch_image(rlr_multi *m, int32_t enable)
{
    if (!m)
        return RLR_E_INVALID;
    return for_each_shard(m, [&](uint32_t g) { return rlr_index_enable_batch_image(m->shard[g], enable); });
}

int32_t rlr_multi_search_topk(rlr_multi *m, const float *queries, uint32_t n_queries, uint32_t k, float guard_eps,
                              uint64_t *rows_out, float *cos_out, uint32_t *n_out)
{
    if (!m || (n_queries && (!queries || !n_out)) || (n_queries && k && (!rows_out || !cos_out)))
        return RLR_E_INVALID;
    const size_t G = m->shard.size();
    if (m->exchange == 1) {
        bool handled = false;
        const auto t0 = std::chrono::steady_clock::now();
        const int32_t xs = search_rccl(m, queries, n_queries, k, guard_eps, rows_out, cos_out, n_out, &handled);
        if (xs != RLR_OK)
            return xs;
        if (handled) {
            m->n_topk_rccl++;
            m->topk_exchange_ns += static_cast<uint64_t>(
                std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
            return RLR_OK;
        }
        m->n_topk_fellback++; // outside the merge kernel's shape, or a shard's guard band overflowed: host merge below
    }
    m->n_topk_host++;
    std::vector<std::vector<uint64_t>> r(G);
    std::vector<std::vector<float>> c(G);
    std::vector<std::vector<uint32_t>> cnt(G);
    const int32_t st = for_each_shard(m, [&](uint32_t g) {
        r[g].assign(static_cast<size_t>(n_queries) * k, 0);
        c[g].assign(static_cast<size_t>(n_queries) * k, 0.0f);
        cnt[g].assign(n_queries, 0);
        return rlr_search_topk(m->shard[g], queries, n_queries, k, guard_eps, r[g].data(), c[g].data(), cnt[g].data());
    });
    if (st != RLR_OK)
        return st;
    // k-way merge of G sorted lists per query (shards are ascending row ranges, so global row order
    // carries every shard's own tie rule)
    std::vector<uint32_t> pos(G);
    for (uint32_t q = 0; q < n_queries; ++q) {
        std::fill(pos.begin(), pos.end(), 0u);
        uint32_t n = 0;
        while (n < k) {
            int best = -1;
            float bs = 0.0f;
            uint64_t br = 0;
            for (size_t g = 0; g < G; ++g) {
                if (pos[g] >= cnt[g][q])
                    continue;
                const size_t i = static_cast<size_t>(q) * k + pos[g];
                const float s = c[g][i];
                const uint64_t row = m->base[g] + r[g][i];
                if (best < 0 || hit_before(s, row, bs, br)) {
                    best = static_cast<int>(g);
                    bs = s;
                    br = row;
                }
            }
            if (best < 0)
                break;
            rows_out[static_cast<size_t>(q) * k + n] = br;
            cos_out[static_cast<size_t>(q) * k + n] = bs;
            pos[best]++;
            n++;
        }
        n_out[q] = n;
    }
    return RLR_OK;
}

int32_t rlr_multi_score_rows(rlr_multi *m, const float *query, const uint64_t *rows, uint32_t n, float *cos_out)
{
    if (!m || (n && (!query || !rows || !cos_out)))
        return RLR_E_INVALID;
    const size_t G = m->shard.size();
    std::vector<std::vector<uint64_t>> local(G);
    std::vector<std::vector<uint32_t>> where(G);
    for (uint32_t i = 0; i < n; ++i) {
        if (rows[i] >= m->n_rows)
            return RLR_E_RANGE;
        uint32_t g;
        uint64_t l;
        shard_of(m, rows[i], &g, &l);
        local[g].push_back(l);
        where[g].push_back(i);
    }
    return for_each_shard(m, [&](uint32_t g) -> int32_t {
        if (local[g].empty())
            return RLR_OK;
        std::vector<float> out(local[g].size());
        const int32_t st = rlr_score_rows(m->shard[g], query, local[g].data(), static_cast<uint32_t>(local[g].size()), out.data());
        if (st != RLR_OK)
            return st;
        for (size_t j = 0; j < out.size(); ++j)
            cos_out[where[g][j]] = out[j];
        return RLR_OK;
    });
}

int32_t rlr_multi_fetch_rows(rlr_multi *m, const uint64_t *rows, uint32_t n, float *out)
{
    if (!m || (n && (!rows || !out)))
        return RLR_E_INVALID;
    const size_t G = m->shard.size();
    std::vector<std::vector<uint64_t>> local(G);
    std::vector<std::vector<uint32_t>> where(G);
    for (uint32_t i = 0; i < n; ++i) {
        if (rows[i] >= m->n_rows)
            return RLR_E_RANGE;
        uint32_t g;
        uint64_t l;
        shard_of(m, rows[i], &g, &l);
        local[g].push_back(l);
        where[g].push_back(i);
    }
    return for_each_shard(m, [&](uint32_t g) -> int32_t {
        if (local[g].empty())
            return RLR_OK;
        std::vector<float> buf(local[g].size() * m->dim);
        const int32_t st = rlr_fetch_rows(m->shard[g], local[g].data(), static_cast<uint32_t>(local[g].size()), buf.data());
        if (st != RLR_OK)
            return st;
        for (size_t j = 0; j < local[g].size(); ++j)
            std::memcpy(out + static_cast<size_t>(where[g][j]) * m->dim, buf.data() + j * m->dim, m->dim * sizeof(float));
        return RLR_OK;
    });
}

int32_t rlr_multi_mmr_select(rlr_multi *m, const uint64_t *pool_rows, const float *pool_scores, uint32_t P, uint32_t k,
                             float lambda, uint32_t *order_out, float *mmr_out, uint32_t *n_out)
{
    if (!m || !n_out)
        return RLR_E_INVALID;
    *n_out = 0;
    if (P == 0)
        return RLR_OK;
    if (!pool_rows || !pool_scores || !order_out)
        return RLR_E_INVALID;
    // winner-row exchange: the pool rows go device to device to the shard that runs this query's MMR
    return multi_mmr(m, pool_rows, pool_scores, &P, 1, P, k, lambda, order_out, mmr_out, n_out);
}

int32_t rlr_multi_mmr_select_batch(rlr_multi *m, const uint64_t *pool_rows, const float *pool_scores, const uint32_t *pool_sizes,
                                   uint32_t n_queries, uint32_t P, uint32_t k, float lambda, uint32_t *order_out, float *mmr_out,
                                   uint32_t *n_out)
{
    if (!m || (n_queries && (!pool_rows || !pool_scores || !pool_sizes || !order_out || !n_out)))
        return RLR_E_INVALID;
    return multi_mmr(m, pool_rows, pool_scores, pool_sizes, n_queries, P, k, lambda, order_out, mmr_out, n_out);
}

int32_t rlr_multi_stats(rlr_multi *m, rlr_multi_stats_t *out, int32_t reset)
{
    if (!m || !out)
        return RLR_E_INVALID;
    out->n_topk_rccl = m->n_topk_rccl.load();
    out->n_topk_host_merge = m->n_topk_host.load();
    out->n_topk_rccl_fell_back = m->n_topk_fellback.load();
    out->topk_rccl_ms = static_cast<double>(m->topk_exchange_ns.load()) * 1e-6;
    out->n_mmr_exchanges = m->n_mmr_exchanges.load();
    out->mmr_exchange_bytes = m->mmr_exchange_bytes.load();
    out->mmr_exchange_ms = static_cast<double>(m->mmr_exchange_ns.load()) * 1e-6;
    out->n_mmr_host_bounces = m->n_mmr_host_bounces.load();
    if (reset) {
        m->n_mmr_host_bounces = 0;
        m->n_topk_rccl = 0;
        m->n_topk_host = 0;
        m->n_topk_fellback = 0;
        m->topk_exchange_ns = 0;
        m->n_mmr_exchanges = 0;
        m->mmr_exchange_bytes = 0;
        m->mmr_exchange_ns = 0;
    }
    return RLR_OK;
}

// ---- RagEngine::search / search_with_diversity over the sharded corpus (include/rlr_engine.h) --------------------------

static int32_t multi_backend(rlr_multi *m, MultiBackend *be)
{
    be->m = m;
    be->n_rows = m->n_rows;
    be->dim = m->dim;
    return RLR_OK;
}

int32_t rlr_multi_engine_search(rlr_multi *m, const float *query_raw, uint32_t dq, uint32_t top_k,
                                const rlr_query_weights *weights, const uint64_t *lex_rows, const float *lex_scores,
                                uint32_t n_lex, int32_t stage, rlr_search_hit *out, uint32_t cap, uint32_t *n_out)
{
    if (!m || !n_out || (!query_raw && dq) || (n_lex && (!lex_rows || !lex_scores)))
        return RLR_E_INVALID;
    *n_out = 0;
    rlr_resolved_weights w;
    rlr_resolve_weights(weights, &w);
    MultiBackend be;
    multi_backend(m, &be);
    std::vector<rlr_host::Cand> res;
    const int32_t st = rlr_host::generic_search(be, query_raw, dq, top_k, w, lex_rows, lex_scores, n_lex, stage, res);
    if (st != RLR_OK)
        return st;
    if (!res.empty() && !out)
        return RLR_E_INVALID;
    rlr_host::emit(res, out, cap, n_out);
    return RLR_OK;
}

int32_t rlr_multi_engine_search_with_diversity(rlr_multi *m, const float *query_raw, uint32_t dq, uint32_t top_k,
                                               float diversity_factor, const rlr_query_weights *weights,
                                               const uint64_t *lex_rows, const float *lex_scores, uint32_t n_lex,
                                               rlr_search_hit *out, uint32_t cap, uint32_t *n_out)
{
    if (!m || !n_out || (!query_raw && dq) || (n_lex && (!lex_rows || !lex_scores)))
        return RLR_E_INVALID;
    *n_out = 0;
    if (diversity_factor < 0.0f) diversity_factor = 0.0f; // f32::clamp(0.0, 1.0) (:725); NaN takes the MMR branch
    if (diversity_factor > 1.0f) diversity_factor = 1.0f;
    rlr_resolved_weights w;
    rlr_resolve_weights(weights, &w);
    MultiBackend be;
    multi_backend(m, &be);
    std::vector<rlr_host::Cand> res;
    const int32_t st = rlr_host::generic_search_with_diversity(be, query_raw, dq, top_k, diversity_factor, w, lex_rows, lex_scores,
                                                               n_lex, res);
    if (st != RLR_OK)
        return st;
    if (!res.empty() && !out)
        return RLR_E_INVALID;
    rlr_host::emit(res, out, cap, n_out);
    return RLR_OK;
}

int32_t rlr_multi_engine_search_text(rlr_multi *m, rlr_lexical *lex, const float *query_raw, uint32_t dq,
                                     const char *query_tokens, size_t tokens_len, uint32_t top_k, float diversity_factor,
                                     int32_t stage, const rlr_query_weights *weights, rlr_search_hit *out, uint32_t cap,
                                     uint32_t *n_out)
{
    if (!m || !lex || !n_out || (!query_raw && dq) || (tokens_len && !query_tokens))
        return RLR_E_INVALID;
    *n_out = 0;
    if (diversity_factor < 0.0f) diversity_factor = 0.0f;
    if (diversity_factor > 1.0f) diversity_factor = 1.0f;
    const bool diversify = !(diversity_factor == 0.0f);
    // the top_k `search` works with (:490, :735) and its lexical limit `top_k.saturating_mul(5)` (:505)
    const uint32_t k_seen = std::max<uint32_t>(diversify ? rlr_host::pool_size_of(top_k) : top_k, 1u);
    const uint32_t limit = static_cast<uint32_t>(std::min<uint64_t>(static_cast<uint64_t>(k_seen) * 5, 0xFFFFFFFFull));
    const uint32_t lcap = limit == 0 ? RLR_LEXICAL_MAX_LIMIT : std::min<uint32_t>(limit, RLR_LEXICAL_MAX_LIMIT);
    std::vector<uint64_t> lrows(lcap);
    std::vector<float> lscores(lcap);
    uint32_t n_lex = 0;
    const int32_t st = rlr_lexical_score(lex, query_tokens, tokens_len, limit, lrows.data(), lscores.data(), &n_lex);
    if (st != RLR_OK)
        return st;
    return diversify ? rlr_multi_engine_search_with_diversity(m, query_raw, dq, top_k, diversity_factor, weights, lrows.data(),
                                                               lscores.data(), n_lex, out, cap, n_out)
                     : rlr_multi_engine_search(m, query_raw, dq, top_k, weights, lrows.data(), lscores.data(), n_lex, stage, out,
                                               cap, n_out);
}

int32_t rlr_multi_engine_search_with_diversity_batch(rlr_multi *m, const float *queries_raw, uint32_t dq, uint32_t n_queries,
                                                     uint32_t top_k, float diversity_factor, const rlr_query_weights *weights,
                                                     rlr_search_hit *out, uint32_t cap, uint32_t *n_out)
{
    if (!m || !n_out || (n_queries && !queries_raw && dq) || (n_queries && cap && !out))
        return RLR_E_INVALID;
    for (uint32_t q = 0; q < n_queries; ++q)
        n_out[q] = 0;
    if (n_queries == 0)
        return RLR_OK;
    if (diversity_factor < 0.0f) diversity_factor = 0.0f;
    if (diversity_factor > 1.0f) diversity_factor = 1.0f;
    rlr_resolved_weights w;
    rlr_resolve_weights(weights, &w);
    MultiBackend be;
    multi_backend(m, &be);
    std::vector<std::vector<rlr_host::Cand>> results;
    const int32_t st = rlr_host::generic_search_with_diversity_batch(be, queries_raw, dq, n_queries, top_k, diversity_factor, w,
                                                                     results);
    if (st != RLR_OK)
        return st;
    for (uint32_t q = 0; q < n_queries; ++q)
        rlr_host::emit(results[q], out + static_cast<size_t>(q) * cap, cap, &n_out[q]);
    return RLR_OK;
}

int32_t rlr_multi_engine_embedding_candidates(rlr_multi *m, const float *query_raw, uint32_t dq, uint32_t count,
                                              uint64_t *rows_out, float *scores_out, uint32_t *n_out)
{
    if (!m || !n_out || (!query_raw && dq))
        return RLR_E_INVALID;
    *n_out = 0;
    if (m->n_rows == 0 || count == 0)
        return RLR_OK;
    const std::vector<float> q = rlr_host::prepare_query(query_raw, dq, m->dim);
    return rlr_multi_search_topk(m, q.data(), 1, count, -1.0f, rows_out, scores_out, n_out);
}

} // extern "C"
