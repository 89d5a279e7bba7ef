// engine_host.h -- the host half of RagEngine::search / search_with_diversity (reference src/rag_engine.rs:470-701,
// :717-759) written against a small "backend" of device entry points, so that one index on one GPU (csrc/engine.cpp,
// include/rlr_engine.h) and a corpus sharded over several GPUs (csrc/multi.cpp, rlr_multi_engine_*) run the SAME blend,
// pool sizing, tie handling and fall-backs.  Host arithmetic is the reference's (built with -ffp-contract=off); no dot
// product over corpus rows is computed on the host.
//
// A backend provides (all rows are GLOBAL row numbers):
//   uint64_t n_rows;  uint32_t dim;
//   int32_t topk(const float *queries, uint32_t nq, uint32_t k, uint64_t *rows, float *cos, uint32_t *n) const;
//        -- per query the k best rows by reference-order cosine, (cos desc, row asc), NaN last
//   int32_t score_rows(const float *query, const uint64_t *rows, uint32_t n, float *cos) const;
//   int32_t mmr(const uint64_t *pool_rows, const float *pool_scores, const uint32_t *pool_sizes, uint32_t nq, uint32_t P,
//               uint32_t k, float lambda, uint32_t *order, uint32_t *n_sel) const;
//        -- mmr_diversify (:767-839) over nq pools strided by P (nq == 1: P up to 4096, else up to 1024)
#pragma once

#include "../../include/rlr_engine.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace rlr_host {

struct Cand {
    uint64_t row;
    float c, e, l;
};

// (combined desc, row asc), NaN last
inline bool cand_before(const Cand &a, const Cand &b)
{
    const bool an = std::isnan(a.c), bn = std::isnan(b.c);
    if (an || bn) {
        if (an != bn)
            return bn;
        return a.row < b.row;
    }
    if (a.c != b.c)
        return a.c > b.c;
    return a.row < b.row;
}

inline float combine(const rlr_resolved_weights &w, float e, float l)
{
    const float t0 = w.embedding * e; // :531-532, two rounded products then one add
    const float t1 = w.lexical * l;
    return t0 + t1;
}

// query_embedding after `normalize` (:494), shaped to the index dim the way dot_product's
// zip would see it (:1778): extra components are dropped, missing ones contribute 0.
inline std::vector<float> prepare_query(const float *query_raw, uint32_t dq, uint32_t dim)
{
    std::vector<float> q(query_raw, query_raw + dq);
    rlr_normalize(q.data(), q.size());
    q.resize(dim, 0.0f);
    return q;
}

// prepare_query for a batch: the same per-query arithmetic (normalize's strict left-to-right f32 sum of squares, then one
// division per element, :1763-1771), eight queries at a time so that eight independent chains are in flight -- one dependent
// chain per query costs 4 cycles per element (1.2 ms for config 5's 1024 x 1024-d batch).  out: nq x dim, zero-extended / cut.
inline void prepare_queries(const float *queries_raw, uint32_t dq, uint32_t nq, uint32_t dim, float *out)
{
    for (uint32_t q0 = 0; q0 < nq; q0 += 8) {
        const uint32_t m = std::min<uint32_t>(8, nq - q0);
        float s[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        const float *v[8];
        for (uint32_t j = 0; j < 8; ++j)
            v[j] = queries_raw + static_cast<size_t>(q0 + (j < m ? j : 0)) * dq;
        for (uint32_t i = 0; i < dq; ++i)
            for (uint32_t j = 0; j < 8; ++j) {
                const float p = v[j][i] * v[j][i];
                s[j] = s[j] + p;
            }
        for (uint32_t j = 0; j < m; ++j) {
            float *o = out + static_cast<size_t>(q0 + j) * dim;
            const uint32_t n = std::min(dq, dim);
            if (s[j] > 1e-20f) {
                const float norm = std::sqrt(s[j]);
                for (uint32_t i = 0; i < n; ++i)
                    o[i] = v[j][i] / norm;
            } else {
                std::memcpy(o, v[j], n * sizeof(float));
            }
            for (uint32_t i = n; i < dim; ++i)
                o[i] = 0.0f;
        }
    }
}

// The lexical map of search() (:505-506, a HashMap: a repeated chunk keeps its LAST score) as ascending unique rows,
// and max_lexical (:515-519: over every pair, floored at f32::EPSILON).
struct LexPrep {
    std::vector<uint64_t> rows;
    std::vector<float> scores;
    float max_lex = 1.1920929e-07f;
    // normalised lexical score of `row`, 0 when it has none (:527-530)
    bool find(uint64_t row, float *l) const
    {
        const auto it = std::lower_bound(rows.begin(), rows.end(), row);
        if (it == rows.end() || *it != row)
            return false;
        *l = scores[static_cast<size_t>(it - rows.begin())] / max_lex;
        return true;
    }
};

inline LexPrep prepare_lexical(uint64_t N, const uint64_t *lex_rows, const float *lex_scores, uint32_t n_lex)
{
    LexPrep p;
    float max_lex = 0.0f;
    std::vector<uint32_t> order;
    order.reserve(n_lex);
    for (uint32_t i = 0; i < n_lex; ++i) {
        max_lex = std::fmax(max_lex, lex_scores[i]);
        if (lex_rows[i] < N)
            order.push_back(i);
    }
    if (max_lex >= 1.1920929e-07f)
        p.max_lex = max_lex;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return lex_rows[a] < lex_rows[b]; });
    p.rows.reserve(order.size());
    p.scores.reserve(order.size());
    for (size_t i = 0; i < order.size(); ++i) {
        if (i + 1 < order.size() && lex_rows[order[i + 1]] == lex_rows[order[i]])
            continue; // a later pair for the same chunk overwrites this one
        p.rows.push_back(lex_rows[order[i]]);
        p.scores.push_back(lex_scores[order[i]]);
    }
    return p;
}


inline void emit(const std::vector<Cand> &v, rlr_search_hit *out, uint32_t cap, uint32_t *n_out)
{
    const uint32_t n = static_cast<uint32_t>(std::min<size_t>(v.size(), cap));
    for (uint32_t i = 0; i < n; ++i) {
        out[i].row = v[i].row;
        out[i].score = v[i].c;
        out[i].embedding_score = v[i].e;
        out[i].lexical_score = v[i].l;
        out[i].initial_score = v[i].c;
    }
    *n_out = n;
}

inline uint32_t pool_size_of(uint32_t top_k) // max(3k, k + 10) (:734)
{
    const uint64_t p3 = static_cast<uint64_t>(top_k) * 3, p10 = static_cast<uint64_t>(top_k) + 10;
    return static_cast<uint32_t>(std::min<uint64_t>(std::max(p3, p10), 0xFFFFFFFFull));
}

// search() from the prepared query on (:496-565, :667-698), exact-scan branch, reranker absent: `need` candidates by
// (combined desc, row asc).  The part of search_impl that needs nothing but topk / score_rows.
template <typename B>
int32_t blend_search(const B &be, const std::vector<float> &q, uint64_t need, const rlr_resolved_weights &w, const LexPrep &lex,
                     std::vector<Cand> &result)
{
    result.clear();
    const uint64_t N = be.n_rows;
    const std::vector<uint64_t> &lrows = lex.rows;
    int32_t st = RLR_OK;
    std::vector<float> lcos(lrows.size());
    if (!lrows.empty()) {
        st = be.score_rows(q.data(), lrows.data(), static_cast<uint32_t>(lrows.size()), lcos.data());
        if (st != RLR_OK)
            return st;
    }
    std::vector<Cand> cands;
    if (w.embedding == 0.0f) {
        // every non-lexical row scores 0*e + w_l*0 = 0 -> they tie and the build's tie rule
        // (row asc) picks the lowest rows; no scan needed, only their cosines for reporting.
        const uint64_t take = std::min<uint64_t>(N, need + lrows.size());
        std::vector<uint64_t> rows(take);
        for (uint64_t r = 0; r < take; ++r)
            rows[r] = r;
        std::vector<float> cosv(take);
        st = be.score_rows(q.data(), rows.data(), static_cast<uint32_t>(take), cosv.data());
        if (st != RLR_OK)
            return st;
        for (uint64_t r = 0; r < take; ++r) {
            float l = 0.0f;
            (void)lex.find(r, &l);
            cands.push_back({r, combine(w, cosv[r], l), cosv[r], l});
        }
        for (size_t i = 0; i < lrows.size(); ++i)
            if (lrows[i] >= take) {
                const float l = lex.scores[i] / lex.max_lex;
                cands.push_back({lrows[i], combine(w, lcos[i], l), lcos[i], l});
            }
        std::sort(cands.begin(), cands.end(), cand_before);
    } else {
        // Non-lexical rows are ordered by cosine alone (w_e > 0 and rounding is monotone), so
        // the device top-(need + n_lex + slack) by cosine, united with the lexical rows,
        // contains the top-`need` by combined score.  Distinct cosines can round to the same
        // combined score; if such a tie chain reaches the last fetched row the fetch is widened.
        uint64_t fetch = std::min<uint64_t>(N, need + lrows.size() + 8);
        std::vector<uint64_t> rows;
        std::vector<float> cosv;
        std::vector<char> seen(lrows.size());
        for (;;) {
            rows.assign(fetch, 0);
            cosv.assign(fetch, 0.0f);
            uint32_t got = 0;
            st = be.topk(q.data(), 1, static_cast<uint32_t>(fetch), rows.data(), cosv.data(), &got);
            if (st != RLR_OK)
                return st;
            cands.clear();
            std::fill(seen.begin(), seen.end(), 0);
            for (uint32_t i = 0; i < got; ++i) {
                float l = 0.0f;
                const auto it = std::lower_bound(lrows.begin(), lrows.end(), rows[i]);
                if (it != lrows.end() && *it == rows[i]) {
                    const size_t j = static_cast<size_t>(it - lrows.begin());
                    l = lex.scores[j] / lex.max_lex;
                    seen[j] = 1;
                }
                cands.push_back({rows[i], combine(w, cosv[i], l), cosv[i], l});
            }
            for (size_t i = 0; i < lrows.size(); ++i)
                if (!seen[i]) {
                    const float l = lex.scores[i] / lex.max_lex;
                    cands.push_back({lrows[i], combine(w, lcos[i], l), lcos[i], l});
                }
            std::sort(cands.begin(), cands.end(), cand_before);
            if (got >= N || got == 0)
                break;
            const float c_tail = combine(w, cosv[got - 1], 0.0f); // bound on every unfetched row
            if (cands.size() >= need && (std::isnan(c_tail) || cands[need - 1].c > c_tail))
                break;
            fetch = std::min<uint64_t>(N, fetch * 2);
        }
    }
    if (cands.size() > need)
        cands.resize(need);
    result.swap(cands);
    return RLR_OK;
}

// how many candidates search(top_k) keeps: the initial_k a reranker would get (stage 1, :544) or the final cut (:667-698)
inline uint64_t need_of(uint64_t N, uint32_t top_k, int32_t stage)
{
    if (top_k < 1) // :490
        top_k = 1;
    const uint64_t want3 = static_cast<uint64_t>(top_k) * 3 > top_k ? static_cast<uint64_t>(top_k) * 3 : top_k;
    const uint64_t initial_k = std::min<uint64_t>(N, want3);
    return stage ? initial_k : std::min<uint64_t>(initial_k, top_k);
}

// RagEngine::search on a backend without fused kernels.
template <typename B>
int32_t generic_search(const B &be, const float *query_raw, uint32_t dq, uint32_t top_k, const rlr_resolved_weights &w,
                       const uint64_t *lex_rows, const float *lex_scores, uint32_t n_lex, int32_t stage, std::vector<Cand> &result)
{
    result.clear();
    if (be.n_rows == 0) // :476-478
        return RLR_OK;
    const std::vector<float> q = prepare_query(query_raw, dq, be.dim);
    const LexPrep lex = prepare_lexical(be.n_rows, lex_rows, lex_scores, n_lex);
    return blend_search(be, q, need_of(be.n_rows, top_k, stage), w, lex, result);
}

// RagEngine::search_with_diversity (:717-759) on a backend without fused kernels; `lambda` already clamped.
template <typename B>
int32_t generic_search_with_diversity(const B &be, const float *query_raw, uint32_t dq, uint32_t top_k, float lambda,
                                      const rlr_resolved_weights &w, const uint64_t *lex_rows, const float *lex_scores,
                                      uint32_t n_lex, std::vector<Cand> &picked)
{
    picked.clear();
    if (lambda == 0.0f) // :728-730
        return generic_search(be, query_raw, dq, top_k, w, lex_rows, lex_scores, n_lex, 0, picked);
    std::vector<Cand> pool;
    int32_t st = generic_search(be, query_raw, dq, pool_size_of(top_k), w, lex_rows, lex_scores, n_lex, 0, pool); // :735
    if (st != RLR_OK || pool.empty())                                                                              // :737-739
        return st;
    const uint32_t P = static_cast<uint32_t>(pool.size());
    std::vector<uint64_t> rows(P);
    std::vector<float> scores(P);
    for (uint32_t i = 0; i < P; ++i) {
        rows[i] = pool[i].row;
        scores[i] = pool[i].c;
    }
    std::vector<uint32_t> order(P);
    uint32_t n_sel = 0;
    st = be.mmr(rows.data(), scores.data(), &P, 1, P, top_k, lambda, order.data(), &n_sel); // :756
    if (st != RLR_OK)
        return st;
    picked.reserve(n_sel);
    for (uint32_t i = 0; i < n_sel; ++i)
        picked.push_back(pool[order[i]]);
    return RLR_OK;
}

// The additive batched entry point (loop of search_with_diversity over the batch, no lexical candidates): one batched
// top-k, per-query pools on the host, one batched MMR; queries whose fetch cannot decide the pool order (rounding ties
// at its boundary) take the single-query path.  results[q] = that query's hits.
template <typename B>
int32_t generic_search_with_diversity_batch(const B &be, const float *queries_raw, uint32_t dq, uint32_t n_queries,
                                            uint32_t top_k, float lambda, const rlr_resolved_weights &w,
                                            std::vector<std::vector<Cand>> &results)
{
    results.assign(n_queries, {});
    const uint64_t N = be.n_rows;
    if (n_queries == 0 || N == 0)
        return RLR_OK;
    const bool plain = lambda == 0.0f;
    const uint32_t k_eff = std::max<uint32_t>(plain ? top_k : pool_size_of(top_k), 1u); // search() treats 0 as 1 (:490)
    const uint64_t need = std::min<uint64_t>(N, k_eff);
    int32_t st = RLR_OK;
    auto single = [&](uint32_t q) -> int32_t { // reference path for one query of the batch
        return generic_search_with_diversity(be, queries_raw + static_cast<size_t>(q) * dq, dq, top_k, lambda, w, nullptr, nullptr,
                                             0, results[q]);
    };
    if (w.embedding == 0.0f || need > 1024) { // degenerate weight / pool beyond the batched MMR: loop
        for (uint32_t q = 0; q < n_queries; ++q)
            if ((st = single(q)) != RLR_OK)
                return st;
        return RLR_OK;
    }
    std::vector<float> qn(static_cast<size_t>(n_queries) * be.dim);
    prepare_queries(queries_raw, dq, n_queries, be.dim, qn.data());
    const uint32_t fetch = static_cast<uint32_t>(std::min<uint64_t>(N, need + 8));
    std::vector<uint64_t> rows(static_cast<size_t>(n_queries) * fetch);
    std::vector<float> cosv(static_cast<size_t>(n_queries) * fetch);
    std::vector<uint32_t> got(n_queries);
    st = be.topk(qn.data(), n_queries, fetch, rows.data(), cosv.data(), got.data());
    if (st != RLR_OK)
        return st;
    const uint32_t P = static_cast<uint32_t>(need);
    std::vector<std::vector<Cand>> pools(n_queries);
    std::vector<uint32_t> redo;
    for (uint32_t q = 0; q < n_queries; ++q) {
        std::vector<Cand> &c = pools[q];
        c.reserve(got[q]);
        for (uint32_t i = 0; i < got[q]; ++i) {
            const float e = cosv[static_cast<size_t>(q) * fetch + i];
            c.push_back({rows[static_cast<size_t>(q) * fetch + i], combine(w, e, 0.0f), e, 0.0f});
        }
        std::sort(c.begin(), c.end(), cand_before);
        // same boundary rule as blend_search: a rounding tie that reaches the last fetched row
        // cannot be resolved from this fetch -> that query takes the single-query path
        if (got[q] < N && got[q] > 0) {
            const float c_tail = combine(w, cosv[static_cast<size_t>(q) * fetch + got[q] - 1], 0.0f);
            if (!(c.size() >= need && (std::isnan(c_tail) || c[need - 1].c > c_tail)))
                redo.push_back(q);
        }
        if (c.size() > need)
            c.resize(need);
    }
    if (plain) {
        for (uint32_t q = 0; q < n_queries; ++q)
            results[q] = pools[q];
    } else {
        std::vector<uint64_t> prow(static_cast<size_t>(n_queries) * P, 0);
        std::vector<float> psc(static_cast<size_t>(n_queries) * P, 0.0f);
        std::vector<uint32_t> psz(n_queries), order(static_cast<size_t>(n_queries) * P), nsel(n_queries);
        for (uint32_t q = 0; q < n_queries; ++q) {
            psz[q] = static_cast<uint32_t>(pools[q].size());
            for (uint32_t i = 0; i < psz[q]; ++i) {
                prow[static_cast<size_t>(q) * P + i] = pools[q][i].row;
                psc[static_cast<size_t>(q) * P + i] = pools[q][i].c;
            }
        }
        st = be.mmr(prow.data(), psc.data(), psz.data(), n_queries, P, top_k, lambda, order.data(), nsel.data());
        if (st != RLR_OK)
            return st;
        for (uint32_t q = 0; q < n_queries; ++q) {
            results[q].clear();
            for (uint32_t i = 0; i < nsel[q]; ++i)
                results[q].push_back(pools[q][order[static_cast<size_t>(q) * P + i]]);
        }
    }
    for (uint32_t q : redo)
        if ((st = single(q)) != RLR_OK)
            return st;
    return RLR_OK;
}

} // namespace rlr_host
