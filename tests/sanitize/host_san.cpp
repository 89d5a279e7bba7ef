// TEST INFRASTRUCTURE -- driver run under -fsanitize=address,undefined (tests/test_host_sanitize_cpu.py): the host logic
// of csrc/engine.cpp and csrc/jsonio.cpp over the oracle-backed stub of the device ABI, compared with the oracle's own
// search / search_with_diversity / blend on random corpora with lexical candidates, ties, NaN rows and weight overrides.
// Exit code 0 = every comparison matched and no sanitizer report was raised (-fno-sanitize-recover aborts on the first).
#include "../../include/rlr_engine.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

extern "C" {
size_t rlr_o_search(const float *rows, size_t n, size_t d, const float *q_raw, size_t dq, size_t top_k, float w_e, float w_l,
                    const uint64_t *lex_rows, const float *lex_scores, size_t n_lex, int normalize_query, int stage,
                    uint64_t *out_rows, float *out_c, float *out_e, float *out_l, size_t cap);
size_t rlr_o_search_with_diversity(const float *rows, size_t n, size_t d, const float *q_raw, size_t dq, size_t top_k,
                                   float diversity, float w_e, float w_l, const uint64_t *lex_rows, const float *lex_scores,
                                   size_t n_lex, int normalize_query, uint64_t *out_rows, float *out_c, float *out_e,
                                   float *out_l, size_t cap);
size_t rlr_o_blend(const uint64_t *cand_rows, const float *cand_initial, size_t n_cand, const uint64_t *rer_rows,
                   const float *rer_relevance, size_t n_rer, size_t top_k, float w_reranker, float w_initial,
                   uint32_t *out_cand, float *out_score, float *out_rer, int32_t *out_has_rer);
void rlr_o_normalize(float *v, size_t n);
float rlr_o_resolve_weight(int has_override, float w, float dflt);
int32_t rlr_index_create(uint32_t dim, int32_t dtype, int32_t device, rlr_index **out);
int32_t rlr_index_destroy(rlr_index *ix);
}

static int g_fail = 0;
#define CHECK(cond, ...)                                                                                        \
    do {                                                                                                        \
        if (!(cond)) {                                                                                          \
            fprintf(stderr, "MISMATCH %s:%d: ", __FILE__, __LINE__);                                            \
            fprintf(stderr, __VA_ARGS__);                                                                       \
            fprintf(stderr, "\n");                                                                              \
            ++g_fail;                                                                                           \
        }                                                                                                       \
    } while (0)

static bool same_bits(float a, float b) { return std::memcmp(&a, &b, 4) == 0; }

int main(int argc, char **argv)
{
    const char *tmp_dir = argc > 1 ? argv[1] : "/tmp";
    std::mt19937 rng(20261004);
    std::normal_distribution<float> gauss;
    // ---- weights + normalize
    for (float w : {0.0f, 0.5f, 1.0f, -0.1f, 1.5f, NAN, INFINITY})
        for (int has : {0, 1})
            CHECK(same_bits(rlr_resolve_weight(has, w, 0.7f), rlr_o_resolve_weight(has, w, 0.7f)), "resolve_weight %g %d", w, has);
    for (size_t n : {size_t(0), size_t(1), size_t(7), size_t(768)}) {
        std::vector<float> a(n), b;
        for (auto &v : a) v = gauss(rng) * 3.0f;
        if (n == 7) std::fill(a.begin(), a.end(), 1e-12f); // below the 1e-20 threshold: left alone
        b = a;
        rlr_normalize(a.data(), n);
        rlr_o_normalize(b.data(), n);
        CHECK(n == 0 || std::memcmp(a.data(), b.data(), n * 4) == 0, "normalize n=%zu", n);
    }
    // ---- engine host logic over the stub device
    for (int rep = 0; rep < 40; ++rep) {
        const uint32_t dim = rep % 3 == 0 ? 64 : 96;
        const uint64_t n = std::vector<uint64_t>{1, 3, 40, 400, 2500}[rep % 5];
        std::vector<float> rows(n * dim);
        for (auto &v : rows) v = gauss(rng);
        for (uint64_t r = 0; r < n; ++r) rlr_o_normalize(rows.data() + r * dim, dim);
        if (n > 10) {
            std::memcpy(rows.data() + 5 * dim, rows.data() + 3 * dim, dim * 4); // exact tie
            std::fill(rows.begin() + 7 * dim, rows.begin() + 8 * dim, 0.0f);    // zero row
            if (rep % 4 == 1) rows[9 * dim] = NAN;
        }
        rlr_index *ix = nullptr;
        rlr_index_create(dim, RLR_F32, 0, &ix);
        rlr_index_upload(ix, rows.data(), n, 0);
        for (int qi = 0; qi < 4; ++qi) {
            std::vector<float> q(dim);
            for (auto &v : q) v = gauss(rng);
            const uint32_t top_k = std::vector<uint32_t>{0, 1, 5, 100}[(rep + qi) % 4];
            const float div = std::vector<float>{0.0f, 0.3f, 1.0f, 0.7f}[qi];
            rlr_query_weights w{};
            float we = 0.7f, wl = 0.3f;
            if ((rep + qi) % 3 == 0) {
                w.has_embedding = 1; w.embedding = we = std::vector<float>{0.0f, 0.2f, 1.0f, 1e-40f}[(rep / 3 + qi) % 4];
                w.has_lexical = 1; w.lexical = wl = 0.9f;
            }
            std::vector<uint64_t> lr;
            std::vector<float> ls;
            const uint32_t n_lex = static_cast<uint32_t>(std::min<uint64_t>(n, (rep + qi) % 6));
            for (uint32_t i = 0; i < n_lex; ++i) {
                lr.push_back((static_cast<uint64_t>(i) * 7919 + rep) % n);
                ls.push_back(std::vector<float>{0.0f, 0.5f, 2.25f, 7.0f}[(i + qi) % 4]);
            }
            // duplicates in the lexical list collapse like a map insert: drop them up front for both sides
            for (size_t i = 0; i < lr.size(); ++i)
                for (size_t j = i + 1; j < lr.size();)
                    if (lr[j] == lr[i]) { lr.erase(lr.begin() + j); ls.erase(ls.begin() + j); } else ++j;
            const uint32_t cap = 3 * std::max(top_k, 1u) + 16;
            std::vector<rlr_search_hit> hits(cap);
            uint32_t got = 0;
            std::vector<uint64_t> orow(cap);
            std::vector<float> oc(cap), oe(cap), ol(cap);
            for (int stage : {0, 1}) {
                int32_t st = rlr_engine_search(ix, q.data(), dim, top_k, &w, lr.data(), ls.data(), static_cast<uint32_t>(lr.size()),
                                               stage, hits.data(), cap, &got);
                const size_t want = rlr_o_search(rows.data(), n, dim, q.data(), dim, top_k, we, wl, lr.data(), ls.data(), lr.size(),
                                                 1, stage, orow.data(), oc.data(), oe.data(), ol.data(), cap);
                CHECK(st == RLR_OK && got == want, "search count rep %d q %d stage %d: %u vs %zu", rep, qi, stage, got, want);
                for (uint32_t i = 0; i < got && i < want; ++i)
                    CHECK(hits[i].row == orow[i] && same_bits(hits[i].score, oc[i]) && same_bits(hits[i].lexical_score, ol[i]),
                          "search rep %d q %d stage %d i %u", rep, qi, stage, i);
            }
            const uint32_t capd = std::max(3 * std::max(top_k, 1u), top_k + 10) + 4;
            hits.resize(capd); orow.resize(capd); oc.resize(capd); oe.resize(capd); ol.resize(capd);
            int32_t st = rlr_engine_search_with_diversity(ix, q.data(), dim, top_k, div, &w, lr.data(), ls.data(),
                                                          static_cast<uint32_t>(lr.size()), hits.data(), capd, &got);
            const size_t want = rlr_o_search_with_diversity(rows.data(), n, dim, q.data(), dim, top_k, div, we, wl, lr.data(),
                                                            ls.data(), lr.size(), 1, orow.data(), oc.data(), oe.data(), ol.data(), capd);
            CHECK(st == RLR_OK && got == want, "diversity count rep %d q %d: %u vs %zu", rep, qi, got, want);
            for (uint32_t i = 0; i < got && i < want; ++i)
                CHECK(hits[i].row == orow[i] && same_bits(hits[i].score, oc[i]), "diversity rep %d q %d i %u", rep, qi, i);
            // reranker blend over the stage-1 candidates
            std::vector<rlr_search_hit> cand(cap);
            uint32_t nc = 0;
            rlr_engine_search(ix, q.data(), dim, std::max(top_k, 1u), &w, nullptr, nullptr, 0, 1, cand.data(), cap, &nc);
            std::vector<uint64_t> rr, crow(nc);
            std::vector<float> rs, cinit(nc);
            for (uint32_t i = 0; i < nc; ++i) { crow[i] = cand[i].row; cinit[i] = cand[i].initial_score; }
            for (uint32_t i = 0; i < nc; i += 2) { rr.push_back(cand[nc - 1 - i].row); rs.push_back(static_cast<float>((i * 37 % 11)) / 11.0f); }
            if (!rr.empty()) { rr.push_back(rr[0]); rs.push_back(0.9f); rr.push_back(n + 5); rs.push_back(0.5f); } // repeat + unknown row
            std::vector<rlr_search_hit> fin(nc + 1);
            std::vector<float> frer(nc + 1), orer(nc + 1), osc(nc + 1);
            std::vector<int32_t> fhas(nc + 1), ohas(nc + 1);
            std::vector<uint32_t> oidx(nc + 1);
            uint32_t nf = 0;
            st = rlr_engine_blend_reranked(cand.data(), nc, rr.data(), rs.data(), static_cast<uint32_t>(rr.size()), std::max(top_k, 1u),
                                           nullptr, fin.data(), frer.data(), fhas.data(), nc + 1, &nf);
            const size_t nb = rlr_o_blend(crow.data(), cinit.data(), nc, rr.data(), rs.data(), rr.size(), std::max(top_k, 1u), 0.7f, 0.3f,
                                          oidx.data(), osc.data(), orer.data(), ohas.data());
            CHECK(st == RLR_OK && nf == nb, "blend count rep %d q %d: %u vs %zu", rep, qi, nf, nb);
            for (uint32_t i = 0; i < nf && i < nb; ++i)
                CHECK(fin[i].row == crow[oidx[i]] && same_bits(fin[i].score, osc[i]) && fhas[i] == ohas[i], "blend rep %d q %d i %u", rep, qi, i);
        }
        rlr_index_destroy(ix);
    }
    // ---- corpus file reader / formatter
    {
        const std::string path = std::string(tmp_dir) + "/san_corpus.json";
        const uint32_t dim = 5, n = 300;
        std::vector<float> rows(n * dim);
        for (auto &v : rows) v = gauss(rng) * (rng() % 7 == 0 ? 1e-30f : 1.0f);
        rows[3] = INFINITY; rows[9] = NAN; rows[11] = -0.0f; rows[12] = 3.4028235e38f; rows[13] = 1e-45f;
        FILE *f = fopen(path.c_str(), "w");
        fprintf(f, "{\n  \"version\": 2,\n  \"model\": \"m\",\n  \"chunks\": {\n");
        std::vector<char> buf(dim * 40 + 64);
        for (uint32_t r = 0; r < n; ++r) {
            const uint64_t len = rlr_json_format_embedding(rows.data() + r * dim, dim, 6, buf.data(), buf.size());
            CHECK(len <= buf.size(), "format size");
            fprintf(f, "    \"id%u\": {\n      \"text\": \"t \\\"embedding\\\": [1] %u\",\n      \"embedding\": %.*s,\n      \"chunk_index\": %u\n    }%s\n",
                    r, r, static_cast<int>(len), buf.data(), r, r + 1 < n ? "," : "");
        }
        fprintf(f, "  },\n  \"needs_reindex\": false\n}");
        fclose(f);
        rlr_json_corpus c;
        int32_t st = rlr_json_load_corpus(path.c_str(), dim, &c);
        CHECK(st == RLR_OK && c.n_rows == n, "load_corpus status %d rows %llu", st, static_cast<unsigned long long>(c.n_rows));
        for (uint32_t i = 0; st == RLR_OK && i < n * dim; ++i) {
            const bool nonfinite = !std::isfinite(rows[i]);
            CHECK(nonfinite ? std::isnan(c.rows[i]) : same_bits(c.rows[i], rows[i]), "row value %u: %g vs %g", i, c.rows[i], rows[i]);
        }
        CHECK(st != RLR_OK || std::strstr(c.meta_json, "\"embedding\": []") != nullptr, "metadata document");
        rlr_json_free_corpus(&c);
        rlr_index *ix = nullptr;
        rlr_index_create(dim, RLR_F32, 0, &ix);
        st = rlr_index_load_json(ix, path.c_str(), 1, &c);
        CHECK(st == RLR_OK && c.rows == nullptr && c.meta_json != nullptr, "index_load_json");
        rlr_json_free_corpus(&c);
        rlr_index_destroy(ix);
        // truncated / malformed files at every prefix length of a small document: an error or a parse, never a crash
        const std::string doc = "{\"chunks\": {\"a\": {\"embedding\": [1.5, -2e3, null], \"t\": \"x\\\"y\"}, \"b\": {}}, \"version\": 2}";
        for (size_t cut = 0; cut <= doc.size(); ++cut) {
            FILE *g = fopen(path.c_str(), "w");
            fwrite(doc.data(), 1, cut, g);
            fclose(g);
            if (rlr_json_load_corpus(path.c_str(), 3, &c) == RLR_OK)
                rlr_json_free_corpus(&c);
        }
        remove(path.c_str());
    }
    if (g_fail) {
        fprintf(stderr, "%d mismatches\n", g_fail);
        return 1;
    }
    printf("host_san ok\n");
    return 0;
}
