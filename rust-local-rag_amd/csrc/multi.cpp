// multi.cpp -- one process driving several GPUs: contiguous row shards, one PERSISTENT host thread per shard
// (created with the handle, parked on a condition variable between calls), and two forms of the exchange step
// of SURVEY.md 8(e):
//   * host merge (default): every shard's k x (row, score) list comes back to the host, k-way merge there;
//   * RCCL (rlr_multi_set_exchange(m, 1)): every shard leaves its packed partial top-k in device memory
//     (rlr_search_topk_device), one ncclAllGather of n_queries x k x 8 bytes per shard over xGMI inside a
//     group call, merge_topk_kernel on the first device (rlr_merge_topk) -- the path BASELINE.json's north_star
//     names, reachable from the C ABI without torch or one-process-per-GPU.  librccl is loaded with dlopen on
//     first use, so the library has no link-time dependency on it.
// Built on the single-index C ABI plus HIP runtime calls for the exchange buffers; no device code here.
#include "../../include/rlr_gpu.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
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
    rlr_index *scratch = nullptr; // f32 pool index on the first device for MMR
    uint64_t n_rows = 0;
    std::vector<Worker *> worker; // shards 1..G-1 (shard 0 runs on the calling thread)
    // RCCL exchange (rlr_multi_set_exchange)
    int32_t exchange = 0;
    std::mutex xmu; // one collective at a time on the communicators
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> xstream;
    std::vector<void *> d_local, d_gath;
    size_t x_cap = 0; // entries (u64) d_local holds per shard
};

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

void exchange_teardown(rlr_multi *m)
{
    for (size_t g = 0; g < m->comm.size(); ++g)
        if (m->comm[g])
            (void)rccl().CommDestroy(m->comm[g]);
    m->comm.clear();
    for (size_t g = 0; g < m->xstream.size(); ++g) {
        (void)hipSetDevice(m->device[g]);
        if (m->xstream[g]) (void)hipStreamDestroy(m->xstream[g]);
        if (g < m->d_local.size() && m->d_local[g]) (void)hipFree(m->d_local[g]);
        if (g < m->d_gath.size() && m->d_gath[g]) (void)hipFree(m->d_gath[g]);
    }
    m->xstream.clear();
    m->d_local.clear();
    m->d_gath.clear();
    m->x_cap = 0;
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

// The RCCL form of the exchange step.  *handled = false: a shard's guard band overflowed (or the shape is outside
// the merge kernel) and the caller falls through to the host merge, which handles everything.
int32_t search_rccl(rlr_multi *m, const float *queries, uint32_t nq, uint32_t k, float guard_eps, uint64_t *rows_out,
                    float *cos_out, uint32_t *n_out, bool *handled)
{
    *handled = false;
    const size_t G = m->shard.size();
    if (G > 16 || static_cast<uint64_t>(G) * k > 8192 || nq == 0 || k == 0)
        return RLR_OK;
    std::lock_guard<std::mutex> xl(m->xmu);
    const size_t per = static_cast<size_t>(nq) * k;
    if (m->x_cap < per) {
        for (size_t g = 0; g < G; ++g) {
            RLR_X_HIP(hipSetDevice(m->device[g]));
            if (m->d_local[g]) (void)hipFree(m->d_local[g]);
            if (m->d_gath[g]) (void)hipFree(m->d_gath[g]);
            m->d_local[g] = m->d_gath[g] = nullptr;
            RLR_X_HIP(rlr::dev_malloc(&m->d_local[g], per * sizeof(uint64_t)));
            RLR_X_HIP(rlr::dev_malloc(&m->d_gath[g], G * per * sizeof(uint64_t)));
        }
        m->x_cap = per;
    }
    // every shard: the whole local pipeline, k packed results per query left in its device memory
    int32_t st = for_each_shard(m, [&](uint32_t g) {
        return rlr_search_topk_device(m->shard[g], queries, nq, k, guard_eps, m->d_local[g], m->xstream[g]);
    });
    if (st != RLR_OK)
        return st;
    // one all-gather per shard inside a group call: G x nq x k x 8 bytes land on every device, rank-major
    RLR_X_NCCL(rccl().GroupStart());
    for (size_t g = 0; g < G; ++g) {
        const ncclResult_t r = rccl().AllGather(m->d_local[g], m->d_gath[g], per, ncclUint64, m->comm[g], m->xstream[g]);
        if (r != ncclSuccess) {
            (void)rccl().GroupEnd();
            return rlr::set_error(RLR_E_HIP, "ncclAllGather: %s", rccl().GetErrorString(r));
        }
    }
    RLR_X_NCCL(rccl().GroupEnd());
    // merge on the first device (a single process needs the answer once); queued behind its all-gather
    st = rlr_merge_topk(m->device[0], m->d_gath[0], static_cast<uint32_t>(G), nq, k, m->base.data(), rows_out, cos_out,
                        n_out, m->xstream[0]);
    for (size_t g = 1; g < G; ++g) { // the other devices' halves of the collective must be done before the buffers are reused
        RLR_X_HIP(hipSetDevice(m->device[g]));
        RLR_X_HIP(hipStreamSynchronize(m->xstream[g]));
    }
    if (st != RLR_OK)
        return st;
    for (uint32_t q = 0; q < nq; ++q)
        if (n_out[q] == 0xFFFFFFFFu)
            return RLR_OK; // overflow marker: host merge redoes the call
    *handled = true;
    return RLR_OK;
}

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
    const int32_t st = rlr_index_create(dim, RLR_F32, device_ids[0], &m->scratch);
    if (st != RLR_OK) {
        rlr_multi_destroy(m);
        return st;
    }
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
    for (rlr_index *ix : m->shard)
        rlr_index_destroy(ix);
    rlr_index_destroy(m->scratch);
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
    if (!rccl().ok)
        return rlr::set_error(RLR_E_NO_DEVICE, "librccl.so could not be loaded: %s", dlerror() ? dlerror() : "symbols missing");
    std::lock_guard<std::mutex> xl(m->xmu);
    if (m->comm.empty()) {
        m->comm.assign(G, nullptr);
        const ncclResult_t r = rccl().CommInitAll(m->comm.data(), static_cast<int>(G), m->device.data());
        if (r != ncclSuccess) {
            m->comm.clear();
            return rlr::set_error(RLR_E_HIP, "ncclCommInitAll: %s", rccl().GetErrorString(r));
        }
        m->xstream.assign(G, nullptr);
        m->d_local.assign(G, nullptr);
        m->d_gath.assign(G, nullptr);
        for (size_t g = 0; g < G; ++g) {
            RLR_X_HIP(hipSetDevice(m->device[g]));
            RLR_X_HIP(hipStreamCreateWithFlags(&m->xstream[g], hipStreamNonBlocking));
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

int32_t rlr_multi_search_topk(rlr_multi *m, const float *queries, uint32_t n_queries, uint32_t k, float guard_eps,
                              uint64_t *rows_out, float *cos_out, uint32_t *n_out)
{
    if (!m || (n_queries && (!queries || !n_out)) || (n_queries && k && (!rows_out || !cos_out)))
        return RLR_E_INVALID;
    const size_t G = m->shard.size();
    if (m->exchange == 1) {
        bool handled = false;
        const int32_t xs = search_rccl(m, queries, n_queries, k, guard_eps, rows_out, cos_out, n_out, &handled);
        if (xs != RLR_OK)
            return xs;
        if (handled)
            return RLR_OK;
    }
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
    if (m->shard.size() == 1)
        return rlr_mmr_select(m->shard[0], pool_rows, pool_scores, P, k, lambda, order_out, mmr_out, n_out);
    // winner-row exchange: P x dim f32 to the device that runs this query's MMR
    std::vector<float> pool(static_cast<size_t>(P) * m->dim);
    int32_t st = rlr_multi_fetch_rows(m, pool_rows, P, pool.data());
    if (st != RLR_OK)
        return st;
    st = rlr_index_upload(m->scratch, pool.data(), P, 0);
    if (st != RLR_OK)
        return st;
    std::vector<uint64_t> ids(P);
    for (uint32_t i = 0; i < P; ++i)
        ids[i] = i;
    return rlr_mmr_select(m->scratch, ids.data(), pool_scores, P, k, lambda, order_out, mmr_out, n_out);
}

} // extern "C"
