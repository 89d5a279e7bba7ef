// engine.cpp -- host-side mirror of RagEngine::search / search_with_diversity /
// get_embedding_candidates (reference src/rag_engine.rs:470-701, :717-759, :415-461),
// orchestrating the device entry points of rlr_gpu.h.  Host arithmetic here is the
// reference's (this file is compiled with -ffp-contract=off; no FMA, no reassociation).
// No dot product over corpus rows is ever computed on the host: cosines come from
// rlr_search_topk / rlr_score_rows, MMR from rlr_mmr_select.
#include "../../include/rlr_engine.h"
#include "../../include/rlr_lexical.h"
#include "engine_host.h"
#include "lexical_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {

constexpr float kDefaultEmbeddingWeight = 0.7f; // rag_engine.rs:1801-1804
constexpr float kDefaultLexicalWeight = 0.3f;
constexpr float kDefaultRerankerWeight = 0.7f;
constexpr float kDefaultInitialWeight = 0.3f;

bool weight_ok(float w)
{
    return std::isfinite(w) && w >= 0.0f && w <= 1.0f;
}

// parse_weight (rag_engine.rs:1813-1819)
float parse_weight(const char *env, float dflt)
{
    const char *s = std::getenv(env);
    if (!s || !*s)
        return dflt;
    char *end = nullptr;
    const float w = std::strtof(s, &end);
    if (end == s || *end != '\0' || !weight_ok(w))
        return dflt;
    return w;
}

rlr_resolved_weights g_defaults;
std::once_flag g_defaults_once;

const rlr_resolved_weights &cached_defaults()
{
    std::call_once(g_defaults_once, [] {
        g_defaults.embedding = parse_weight("RAG_EMBEDDING_WEIGHT", kDefaultEmbeddingWeight);
        g_defaults.lexical = parse_weight("RAG_LEXICAL_WEIGHT", kDefaultLexicalWeight);
        g_defaults.reranker = parse_weight("RAG_RERANKER_WEIGHT", kDefaultRerankerWeight);
        g_defaults.initial = parse_weight("RAG_INITIAL_SCORE_WEIGHT", kDefaultInitialWeight);
    });
    return g_defaults;
}

using namespace rlr_host;

// one index on one GPU as a backend of engine_host.h
struct SingleBackend {
    rlr_index *idx;
    uint64_t n_rows = 0;
    uint32_t dim = 0;
    int32_t topk(const float *queries, uint32_t nq, uint32_t k, uint64_t *rows, float *cos, uint32_t *n) const
    {
        return rlr_search_topk(idx, queries, nq, k, -1.0f, rows, cos, n);
    }
    int32_t score_rows(const float *query, const uint64_t *rows, uint32_t n, float *cos) const
    {
        return rlr_score_rows(idx, query, rows, n, cos);
    }
    int32_t mmr(const uint64_t *pool_rows, const float *pool_scores, const uint32_t *pool_sizes, uint32_t nq, uint32_t P,
                uint32_t k, float lambda, uint32_t *order, uint32_t *n_sel) const
    {
        if (nq == 1)
            return rlr_mmr_select(idx, pool_rows, pool_scores, pool_sizes[0], k, lambda, order, nullptr, n_sel);
        return rlr_mmr_select_batch(idx, pool_rows, pool_scores, pool_sizes, nq, P, k, lambda, order, nullptr, n_sel);
    }
};

int32_t single_backend(rlr_index *idx, SingleBackend *be)
{
    be->idx = idx;
    return rlr_index_info(idx, &be->n_rows, &be->dim, nullptr, nullptr);
}

int32_t search_impl(rlr_index *idx, const float *query_raw, uint32_t dq, uint32_t top_k,
                    const rlr_resolved_weights &w, const uint64_t *lex_rows, const float *lex_scores,
                    uint32_t n_lex, int32_t stage, std::vector<Cand> &result)
{
    result.clear();
    uint64_t N = 0;
    uint32_t dim = 0;
    int32_t st = rlr_index_info(idx, &N, &dim, nullptr, nullptr);
    if (st != RLR_OK)
        return st;
    if (N == 0) // :476-478
        return RLR_OK;
    if (top_k < 1) // :490
        top_k = 1;
    const std::vector<float> q = prepare_query(query_raw, dq, dim);
    const LexPrep lex = prepare_lexical(N, lex_rows, lex_scores, n_lex);
    const std::vector<uint64_t> &lrows = lex.rows;

    const uint64_t want3 = static_cast<uint64_t>(top_k) * 3 > top_k ? static_cast<uint64_t>(top_k) * 3 : top_k;
    const uint64_t initial_k = std::min<uint64_t>(N, want3);                              // :544
    const uint64_t need = stage ? initial_k : std::min<uint64_t>(initial_k, top_k);       // :667-698

    if (w.embedding > 0.0f && !lrows.empty() && need <= 1024) {
        // lexical candidates: scan -> blend -> order -> cut in one enqueue on the device (rlr_search_hybrid); it hands
        // the query back (fallback) when its fetch cannot decide the order or the sizes exceed its kernels
        const uint32_t nd = static_cast<uint32_t>(need);
        std::vector<uint64_t> rows(nd);
        std::vector<float> cosv(nd), sc(nd), lx(nd);
        uint32_t got = 0;
        int32_t fb = 0;
        st = rlr_search_hybrid(idx, q.data(), nd, 0, 0.0f, 0, w.embedding, w.lexical, lrows.data(), lex.scores.data(),
                               static_cast<uint32_t>(lrows.size()), lex.max_lex, -1.0f, rows.data(), cosv.data(), sc.data(),
                               lx.data(), &got, &fb);
        if (st != RLR_OK)
            return st;
        if (!fb) {
            result.resize(got);
            for (uint32_t i = 0; i < got; ++i)
                result[i] = {rows[i], sc[i], cosv[i], lx[i]};
            return RLR_OK;
        }
    }

    SingleBackend be;
    be.idx = idx;
    be.n_rows = N;
    be.dim = dim;
    return blend_search(be, q, need, w, lex, result);
}

} // namespace

extern "C" {

float rlr_resolve_weight(int32_t has_override, float w, float dflt)
{
    return (has_override && weight_ok(w)) ? w : dflt;
}

void rlr_resolve_weights(const rlr_query_weights *w, rlr_resolved_weights *out)
{
    const rlr_resolved_weights &d = cached_defaults();
    out->embedding = rlr_resolve_weight(w ? w->has_embedding : 0, w ? w->embedding : 0.0f, d.embedding);
    out->lexical = rlr_resolve_weight(w ? w->has_lexical : 0, w ? w->lexical : 0.0f, d.lexical);
    out->reranker = rlr_resolve_weight(w ? w->has_reranker : 0, w ? w->reranker : 0.0f, d.reranker);
    out->initial = rlr_resolve_weight(w ? w->has_initial : 0, w ? w->initial : 0.0f, d.initial);
}

void rlr_normalize(float *v, size_t n)
{
    float norm_sq = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        const float p = v[i] * v[i];
        norm_sq = norm_sq + p;
    }
    if (norm_sq > 1e-20f) {
        const float norm = std::sqrt(norm_sq);
        for (size_t i = 0; i < n; ++i)
            v[i] = v[i] / norm;
    }
}

int32_t rlr_engine_search(rlr_index *idx, const float *query_raw, uint32_t dq, uint32_t top_k,
                          const rlr_query_weights *weights, const uint64_t *lex_rows, const float *lex_scores,
                          uint32_t n_lex, int32_t stage, rlr_search_hit *out, uint32_t cap, uint32_t *n_out)
{
    if (!idx || !n_out || (!query_raw && dq) || (n_lex && (!lex_rows || !lex_scores)))
        return RLR_E_INVALID;
    *n_out = 0;
    rlr_resolved_weights w;
    rlr_resolve_weights(weights, &w);
    std::vector<Cand> res;
    const int32_t st = search_impl(idx, query_raw, dq, top_k, w, lex_rows, lex_scores, n_lex, stage, res);
    if (st != RLR_OK)
        return st;
    if (!res.empty() && !out)
        return RLR_E_INVALID;
    emit(res, out, cap, n_out);
    return RLR_OK;
}

int32_t rlr_engine_search_with_diversity(rlr_index *idx, const float *query_raw, uint32_t dq, uint32_t top_k,
                                         float diversity_factor, const rlr_query_weights *weights,
                                         const uint64_t *lex_rows, const float *lex_scores, uint32_t n_lex,
                                         rlr_search_hit *out, uint32_t cap, uint32_t *n_out)
{
    if (!idx || !n_out || (!query_raw && dq) || (n_lex && (!lex_rows || !lex_scores)))
        return RLR_E_INVALID;
    *n_out = 0;
    // f32::clamp(0.0, 1.0) (:725) -- NaN passes through and takes the MMR branch
    if (diversity_factor < 0.0f) diversity_factor = 0.0f;
    if (diversity_factor > 1.0f) diversity_factor = 1.0f;
    rlr_resolved_weights w;
    rlr_resolve_weights(weights, &w);
    std::vector<Cand> pool;
    if (diversity_factor == 0.0f) { // :728-730
        const int32_t st = search_impl(idx, query_raw, dq, top_k, w, lex_rows, lex_scores, n_lex, 0, pool);
        if (st != RLR_OK)
            return st;
        if (!pool.empty() && !out)
            return RLR_E_INVALID;
        emit(pool, out, cap, n_out);
        return RLR_OK;
    }
    const uint64_t p3 = static_cast<uint64_t>(top_k) * 3, p10 = static_cast<uint64_t>(top_k) + 10;
    const uint32_t pool_size = static_cast<uint32_t>(std::min<uint64_t>(std::max(p3, p10), 0xFFFFFFFFull)); // :734
    if (n_lex == 0 && w.embedding > 0.0f && pool_size <= 1024) {
        // no lexical candidates: search(pool) -> mmr_diversify entirely on the device, one synchronisation
        // (rlr_search_diverse); it hands the query back when its fetch cannot decide the pool order
        uint64_t N = 0;
        uint32_t dim = 0;
        int32_t st0 = rlr_index_info(idx, &N, &dim, nullptr, nullptr);
        if (st0 != RLR_OK)
            return st0;
        if (N == 0)
            return RLR_OK;
        const std::vector<float> q = prepare_query(query_raw, dq, dim);
        const uint32_t kk = static_cast<uint32_t>(std::min<uint64_t>(std::max<uint32_t>(top_k, 1u), pool_size));
        std::vector<uint64_t> rows(kk);
        std::vector<float> cosv(kk), sc(kk);
        uint32_t n_sel = 0;
        int32_t fb = 0;
        st0 = rlr_search_diverse(idx, q.data(), pool_size, top_k, diversity_factor, w.embedding, w.lexical, -1.0f, rows.data(),
                                 cosv.data(), sc.data(), &n_sel, &fb);
        if (st0 != RLR_OK)
            return st0;
        if (!fb) {
            if (n_sel && !out)
                return RLR_E_INVALID;
            std::vector<Cand> picked(n_sel);
            for (uint32_t i = 0; i < n_sel; ++i)
                picked[i] = {rows[i], sc[i], cosv[i], 0.0f};
            emit(picked, out, cap, n_out);
            return RLR_OK;
        }
    }
    if (n_lex > 0 && w.embedding > 0.0f && pool_size <= 1024) {
        // lexical candidates (the usual case: search() scores the query text, :505): the blended pool and the MMR picks
        // come from one enqueue as well (rlr_search_hybrid, diversify = 1)
        uint64_t N = 0;
        uint32_t dim = 0;
        int32_t st0 = rlr_index_info(idx, &N, &dim, nullptr, nullptr);
        if (st0 != RLR_OK)
            return st0;
        if (N == 0)
            return RLR_OK;
        const LexPrep lex = prepare_lexical(N, lex_rows, lex_scores, n_lex);
        if (!lex.rows.empty()) {
            const std::vector<float> q = prepare_query(query_raw, dq, dim);
            const uint32_t kk = static_cast<uint32_t>(std::min<uint64_t>(std::max<uint32_t>(top_k, 1u), pool_size));
            std::vector<uint64_t> rows(kk);
            std::vector<float> cosv(kk), sc(kk), lx(kk);
            uint32_t n_sel = 0;
            int32_t fb = 0;
            st0 = rlr_search_hybrid(idx, q.data(), pool_size, top_k, diversity_factor, 1, w.embedding, w.lexical, lex.rows.data(),
                                    lex.scores.data(), static_cast<uint32_t>(lex.rows.size()), lex.max_lex, -1.0f, rows.data(),
                                    cosv.data(), sc.data(), lx.data(), &n_sel, &fb);
            if (st0 != RLR_OK)
                return st0;
            if (!fb) {
                if (n_sel && !out)
                    return RLR_E_INVALID;
                std::vector<Cand> picked(n_sel);
                for (uint32_t i = 0; i < n_sel; ++i)
                    picked[i] = {rows[i], sc[i], cosv[i], lx[i]};
                emit(picked, out, cap, n_out);
                return RLR_OK;
            }
        }
    }
    int32_t st = search_impl(idx, query_raw, dq, pool_size, w, lex_rows, lex_scores, n_lex, 0, pool); // :735
    if (st != RLR_OK)
        return st;
    if (pool.empty()) // :737-739
        return RLR_OK;
    const uint32_t P = static_cast<uint32_t>(pool.size());
    std::vector<uint64_t> rows(P);
    std::vector<float> scores(P);
    for (uint32_t i = 0; i < P; ++i) {
        rows[i] = pool[i].row;
        scores[i] = pool[i].c;
    }
    std::vector<uint32_t> order(P);
    uint32_t n_sel = 0;
    st = rlr_mmr_select(idx, rows.data(), scores.data(), P, top_k, diversity_factor, order.data(), nullptr, &n_sel); // :756
    if (st != RLR_OK)
        return st;
    std::vector<Cand> picked;
    picked.reserve(n_sel);
    for (uint32_t i = 0; i < n_sel; ++i)
        picked.push_back(pool[order[i]]);
    if (!picked.empty() && !out)
        return RLR_E_INVALID;
    emit(picked, out, cap, n_out);
    return RLR_OK;
}

int32_t rlr_engine_search_text(rlr_index *idx, rlr_lexical *lex, const float *query_raw, uint32_t dq, const char *query_tokens,
                               size_t tokens_len, uint32_t top_k, float diversity_factor, int32_t stage,
                               const rlr_query_weights *weights, rlr_search_hit *out, uint32_t cap, uint32_t *n_out)
{
    if (!idx || !lex || !n_out || (!query_raw && dq) || (tokens_len && !query_tokens))
        return RLR_E_INVALID;
    *n_out = 0;
    if (diversity_factor < 0.0f) diversity_factor = 0.0f; // f32::clamp(0.0, 1.0) (:725); NaN takes the MMR branch
    if (diversity_factor > 1.0f) diversity_factor = 1.0f;
    const bool diversify = !(diversity_factor == 0.0f);
    rlr_resolved_weights w;
    rlr_resolve_weights(weights, &w);
    const uint64_t p3 = static_cast<uint64_t>(top_k) * 3, p10 = static_cast<uint64_t>(top_k) + 10;
    const uint32_t pool_size = static_cast<uint32_t>(std::min<uint64_t>(std::max(p3, p10), 0xFFFFFFFFull)); // :734
    const uint32_t k_seen = std::max<uint32_t>(diversify ? pool_size : top_k, 1u); // the top_k `search` works with (:490)
    const uint32_t limit = static_cast<uint32_t>(std::min<uint64_t>(static_cast<uint64_t>(k_seen) * 5, 0xFFFFFFFFull)); // :505

    uint64_t N = 0;
    uint32_t dim = 0;
    int32_t st = rlr_index_info(idx, &N, &dim, nullptr, nullptr);
    if (st != RLR_OK || N == 0) // :476-478
        return st;
    const uint64_t want3 = static_cast<uint64_t>(k_seen) * 3;
    const uint64_t initial_k = std::min<uint64_t>(N, want3);                                            // :544
    const uint64_t need = (diversify || !stage) ? std::min<uint64_t>(initial_k, k_seen) : initial_k;    // :667-698
    bool retry_exact = false;
    if (w.embedding > 0.0f && need <= 1024) {
        // 1. the cosine scan .. sort goes onto the index' stream; 2. the BM25 kernels onto the lexical index' own stream
        // -- the device runs them side by side, and the host's launch calls for the second batch overlap the scan;
        // 3. blend .. results behind an event join.  One host synchronisation.
        const std::vector<float> q = prepare_query(query_raw, dq, dim);
        const uint32_t nd = static_cast<uint32_t>(need);
        rlr::HybridTicket *ticket = nullptr;
        int32_t fb = 0;
        bool sampled_gave_up = false;
        struct BesideScan { // the BM25 chain goes onto its stream as soon as the scan is queued (search_hybrid_begin calls back)
            rlr_lexical *lex;
            const char *tokens;
            size_t len;
            uint32_t limit;
            rlr::LexPending lp;
            bool queued = false;
        } beside{lex, query_tokens, tokens_len, limit, {}, false};
        st = rlr::search_hybrid_begin(
            idx, q.data(), nd, top_k, diversity_factor, diversify ? 1 : 0, w.embedding, w.lexical,
            std::min<uint32_t>(limit, RLR_LEXICAL_MAX_LIMIT), -1.0f, &ticket, &fb,
            [](void *a, const rlr::LexSink *sink) -> int32_t {
                BesideScan *b = static_cast<BesideScan *>(a);
                const int32_t e = rlr::lexical_enqueue(b->lex, b->tokens, b->len, b->limit, &b->lp, /*need_sorted=*/false,
                                                       /*exact_passes=*/false, sink);
                b->queued = e == RLR_OK;
                return e;
            },
            &beside);
        if (st != RLR_OK) {
            if (beside.queued) // the index side failed behind it (begin drained its own stream): hand the workspace back
                rlr::lexical_finish(&beside.lp, false);
            return st;
        }
        if (fb && beside.queued) {
            // begin handed the query back AFTER the BM25 chain was queued behind the scan (today it only does so before):
            // the workspace and the readers' lock it holds must go back whichever way begin ends
            rlr::lexical_finish(&beside.lp, false);
            beside.queued = false;
        }
        if (!fb) {
            rlr::LexPending &lp = beside.lp;
            const uint32_t n_res = diversify ? static_cast<uint32_t>(std::min<uint64_t>(std::max<uint32_t>(top_k, 1u), nd)) : nd;
            std::vector<uint64_t> rows(n_res);
            std::vector<float> cosv(n_res), sc(n_res), lx(n_res);
            uint32_t got = 0;
            st = rlr::search_hybrid_finish(ticket, &lp, rows.data(), cosv.data(), sc.data(), lx.data(), &got, &fb);
            rlr::lexical_finish(&lp, st == RLR_OK);
            if (st != RLR_OK)
                return st;
            if (!fb) {
                if (got && !out)
                    return RLR_E_INVALID;
                std::vector<Cand> res(got);
                for (uint32_t i = 0; i < got; ++i)
                    res[i] = {rows[i], sc[i], cosv[i], lx[i]};
                emit(res, out, cap, n_out);
                return RLR_OK;
            }
            sampled_gave_up = fb == 3;
        }
        retry_exact = sampled_gave_up;
    }
    // not covered by the fused kernels (or handed back): the pairs on the host, then the entry points that take them
    const uint32_t lcap = limit == 0 ? RLR_LEXICAL_MAX_LIMIT : std::min<uint32_t>(limit, RLR_LEXICAL_MAX_LIMIT);
    std::vector<uint64_t> lrows(lcap);
    std::vector<float> lscores(lcap);
    uint32_t n_lex = 0;
    // (handed back by the BM25 selection's sample, status 3: counted as a retry, straight to the exact passes)
    st = retry_exact ? rlr::lexical_score_exact(lex, query_tokens, tokens_len, limit, lrows.data(), lscores.data(), &n_lex)
                     : rlr_lexical_score(lex, query_tokens, tokens_len, limit, lrows.data(), lscores.data(), &n_lex);
    if (st != RLR_OK)
        return st;
    return diversify ? rlr_engine_search_with_diversity(idx, query_raw, dq, top_k, diversity_factor, weights, lrows.data(),
                                                         lscores.data(), n_lex, out, cap, n_out)
                     : rlr_engine_search(idx, query_raw, dq, top_k, weights, lrows.data(), lscores.data(), n_lex, stage, out, cap,
                                         n_out);
}

int32_t rlr_engine_search_with_diversity_batch(rlr_index *idx, const float *queries_raw, uint32_t dq, uint32_t n_queries,
                                               uint32_t top_k, float diversity_factor, const rlr_query_weights *weights,
                                               rlr_search_hit *out, uint32_t cap, uint32_t *n_out)
{
    if (!idx || !n_out || (n_queries && !queries_raw && dq) || (n_queries && cap && !out))
        return RLR_E_INVALID;
    for (uint32_t q = 0; q < n_queries; ++q)
        n_out[q] = 0;
    if (n_queries == 0)
        return RLR_OK;
    SingleBackend be;
    int32_t st = single_backend(idx, &be);
    if (st != RLR_OK)
        return st;
    if (diversity_factor < 0.0f) diversity_factor = 0.0f;
    if (diversity_factor > 1.0f) diversity_factor = 1.0f;
    rlr_resolved_weights w;
    rlr_resolve_weights(weights, &w);
    std::vector<std::vector<Cand>> results;
    st = generic_search_with_diversity_batch(be, queries_raw, dq, n_queries, top_k, diversity_factor, w, results);
    if (st != RLR_OK)
        return st;
    for (uint32_t q = 0; q < n_queries; ++q)
        emit(results[q], out + static_cast<size_t>(q) * cap, cap, &n_out[q]);
    return RLR_OK;
}

int32_t rlr_engine_blend_reranked(const rlr_search_hit *candidates, uint32_t n_candidates, const uint64_t *rer_rows,
                                  const float *rer_relevance, uint32_t n_reranked, uint32_t top_k,
                                  const rlr_query_weights *weights, rlr_search_hit *out, float *reranker_score_out,
                                  int32_t *has_reranker_out, uint32_t cap, uint32_t *n_out)
{
    if (!n_out || (n_candidates && !candidates) || (n_reranked && (!rer_rows || !rer_relevance)))
        return RLR_E_INVALID;
    *n_out = 0;
    rlr_resolved_weights w;
    rlr_resolve_weights(weights, &w);
    struct Item {
        float score;
        uint32_t cand;
        float rer;
        int32_t has;
    };
    // `b.score.partial_cmp(&a.score).unwrap_or(Equal)` under a stable sort: NaN ties with everything
    auto before = [](const Item &a, const Item &b) { return a.score > b.score; };
    std::vector<Item> res;
    std::vector<char> seen(n_candidates, 0);
    if (n_reranked > 0) { // :602
        float max_rer = 0.0f, max_init = 0.0f;
        for (uint32_t i = 0; i < n_reranked; ++i) max_rer = std::fmax(max_rer, rer_relevance[i]);
        if (!(max_rer >= 1.1920929e-07f)) max_rer = 1.1920929e-07f;
        for (uint32_t i = 0; i < n_candidates; ++i) max_init = std::fmax(max_init, candidates[i].initial_score);
        if (!(max_init >= 1.1920929e-07f)) max_init = 1.1920929e-07f;
        std::unordered_map<uint64_t, uint32_t> where;
        for (uint32_t i = n_candidates; i-- > 0;)
            where[candidates[i].row] = i; // first occurrence wins
        for (uint32_t i = 0; i < n_reranked; ++i) { // :616-654
            auto it = where.find(rer_rows[i]);
            if (it == where.end() || seen[it->second])
                continue;
            seen[it->second] = 1;
            const float rn = rer_relevance[i] / max_rer;
            const float in = candidates[it->second].initial_score / max_init;
            const float t0 = w.reranker * rn;
            const float t1 = w.initial * in;
            res.push_back({t0 + t1, it->second, rer_relevance[i], 1});
        }
        std::stable_sort(res.begin(), res.end(), before); // :657-661
        if (res.size() > top_k)
            res.resize(top_k); // :664
    }
    if (res.size() < top_k) { // :667-698
        std::vector<Item> fb;
        fb.reserve(n_candidates);
        for (uint32_t i = 0; i < n_candidates; ++i)
            fb.push_back({candidates[i].initial_score, i, 0.0f, 0});
        std::stable_sort(fb.begin(), fb.end(), before);
        for (const Item &f : fb) {
            if (res.size() >= top_k)
                break;
            if (!seen[f.cand]) {
                seen[f.cand] = 1;
                res.push_back(f);
            }
        }
    }
    const uint32_t n = static_cast<uint32_t>(std::min<size_t>(res.size(), cap));
    if (n && (!out || !reranker_score_out || !has_reranker_out))
        return RLR_E_INVALID;
    for (uint32_t i = 0; i < n; ++i) {
        out[i] = candidates[res[i].cand];
        out[i].score = res[i].score;
        reranker_score_out[i] = res[i].rer;
        has_reranker_out[i] = res[i].has;
    }
    *n_out = n;
    return RLR_OK;
}

int32_t rlr_engine_embedding_candidates(rlr_index *idx, const float *query_raw, uint32_t dq, uint32_t count,
                                        uint64_t *rows_out, float *scores_out, uint32_t *n_out)
{
    if (!idx || !n_out || (!query_raw && dq))
        return RLR_E_INVALID;
    *n_out = 0;
    uint64_t N = 0;
    uint32_t dim = 0;
    int32_t st = rlr_index_info(idx, &N, &dim, nullptr, nullptr);
    if (st != RLR_OK)
        return st;
    if (N == 0 || count == 0)
        return RLR_OK;
    const std::vector<float> q = prepare_query(query_raw, dq, dim);
    return rlr_search_topk(idx, q.data(), 1, count, -1.0f, rows_out, scores_out, n_out);
}

} // extern "C"
