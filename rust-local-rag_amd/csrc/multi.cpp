// multi.cpp -- one process driving several GPUs: contiguous row shards, one host thread per shard
// per call, host-side merge of the per-shard top-k lists (include/rlr_gpu.h, "one process, several
// GPUs").  Built only on the single-index C ABI; no device code here.
#include "../../include/rlr_gpu.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace rlr {
int32_t set_error(int32_t code, const char *fmt, ...); // index.hip: the calling thread's rlr_last_error() text
}

struct rlr_multi {
    uint32_t dim = 0;
    int32_t dtype = RLR_F32;
    std::vector<rlr_index *> shard;
    std::vector<uint64_t> base; // first global row of each shard (+ total at the end)
    rlr_index *scratch = nullptr; // f32 pool index on the first device for MMR
    uint64_t n_rows = 0;
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

// run f(g) for every shard on its own thread; returns the first failing status.  rlr_last_error() is per thread,
// so a worker's message is carried back to the calling thread (prefixed with the shard it came from).
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
    std::vector<std::thread> th;
    th.reserve(G);
    for (size_t g = 0; g < G; ++g)
        th.emplace_back([&, g] {
            st[g] = f(static_cast<uint32_t>(g));
            if (st[g] != RLR_OK)
                why[g] = rlr_last_error();
        });
    for (auto &t : th)
        t.join();
    for (size_t g = 0; g < G; ++g)
        if (st[g] != RLR_OK)
            return rlr::set_error(st[g], "shard %zu: %s", g, why[g].c_str());
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
    for (rlr_index *ix : m->shard)
        rlr_index_destroy(ix);
    rlr_index_destroy(m->scratch);
    delete m;
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
