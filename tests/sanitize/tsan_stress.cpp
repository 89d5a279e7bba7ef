// TEST INFRASTRUCTURE -- the concurrency contract of SURVEY.md section 8(b) ("multiple searches may call the FFI
// concurrently from different OS threads, never concurrently with a mutation"; src/mcp_server.rs:89, :377 read guards,
// worker.rs:397-399 write lock) driven against the library's HOST code under ThreadSanitizer: the whole of csrc/ compiled
// with -fsanitize=thread and linked against tests/sanitize/stub_hip.cpp instead of libamdhip64.  More caller threads
// than lexical workspaces (8) and than index contexts (RLR_MAX_CONTEXTS) hammer rlr_lexical_score,
// rlr_engine_search_text, rlr_engine_search_with_diversity, rlr_search_topk and rlr_mmr_select; between rounds the main
// thread mutates both indexes (append + add_chunk, delete + remove_rows) with no search in flight, as the reference's
// write lock guarantees.  Kernels do nothing on the stub, so answers are not checked here (the GPU suite does that): the
// pass criterion is "no ThreadSanitizer report, no deadlock, every call returns".
#include "../../include/rlr_engine.h"
#include "../../include/rlr_lexical.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

namespace {
uint64_t g_seed = 0x9E3779B97F4A7C15ull;
uint64_t next(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
std::string words(uint64_t *s, int n)
{
    std::string out;
    for (int i = 0; i < n; ++i) {
        char b[16];
        // a skewed vocabulary: low ids are common
        const uint64_t r = next(s);
        const unsigned id = static_cast<unsigned>((r % 1000) * (r >> 40) % 1000 / 7);
        snprintf(b, sizeof b, "t%04u", id);
        if (i)
            out += ' ';
        out += b;
    }
    return out;
}
std::atomic<int> g_fail{0};
std::atomic<bool> g_said[2];
#define CHECK(call)                                                                                                    \
    do {                                                                                                               \
        const int32_t st_ = (call);                                                                                    \
        if (st_ != RLR_OK) {                                                                                           \
            fprintf(stderr, "%s -> %d (%s)\n", #call, st_, rlr_last_error());                                           \
            g_fail++;                                                                                                  \
        }                                                                                                              \
    } while (0)
} // namespace

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 3;
    const int iters = argc > 2 ? atoi(argv[2]) : 40;
    const uint32_t dim = 64;
    uint64_t n_rows = 6000;
    rlr_index *ix = nullptr;
    rlr_lexical *lx = nullptr;
    CHECK(rlr_index_create(dim, RLR_F32, 0, &ix));
    CHECK(rlr_lexical_create(0, &lx));
    std::vector<float> rows(n_rows * dim);
    for (auto &v : rows)
        v = static_cast<float>(static_cast<int64_t>(next(&g_seed) % 2001) - 1000) / 1000.0f;
    CHECK(rlr_index_upload(ix, rows.data(), n_rows, 1));
    for (uint64_t r = 0; r < n_rows; ++r) {
        const std::string t = words(&g_seed, 20);
        CHECK(rlr_lexical_add_chunk(lx, r, t.data(), t.size()));
    }
    const int thread_counts[] = {4, 12, 16, 32};
    for (int rd = 0; rd < rounds; ++rd) {
        for (int nt : thread_counts) {
            std::vector<std::thread> ts;
            for (int t = 0; t < nt; ++t)
                ts.emplace_back([&, t] {
                    uint64_t s = 1234567 + 977 * t + 31 * rd;
                    std::vector<float> q(dim);
                    std::vector<rlr_search_hit> hits(400);
                    std::vector<uint64_t> lrows(RLR_LEXICAL_MAX_LIMIT), prow(64);
                    std::vector<float> lsc(RLR_LEXICAL_MAX_LIMIT), cosv(400), psc(64), mmr(64);
                    std::vector<uint32_t> order(64);
                    for (int it = 0; it < iters; ++it) {
                        for (auto &v : q)
                            v = static_cast<float>(static_cast<int64_t>(next(&s) % 2001) - 1000) / 1000.0f;
                        const std::string text = words(&s, 1 + static_cast<int>(next(&s) % 5));
                        uint32_t n = 0;
                        switch ((t + it) % 5) {
                        case 0:
                            CHECK(rlr_lexical_score(lx, text.data(), text.size(), 500, lrows.data(), lsc.data(), &n));
                            break;
                        case 1:
                            CHECK(rlr_engine_search_text(ix, lx, q.data(), dim, text.data(), text.size(), 20, 0.3f, 0, nullptr,
                                                         hits.data(), 400, &n));
                            break;
                        case 2:
                            CHECK(rlr_engine_search_with_diversity(ix, q.data(), dim, 20, 0.3f, nullptr, nullptr, nullptr, 0,
                                                                   hits.data(), 400, &n));
                            break;
                        case 3: {
                            std::vector<uint32_t> nout(1);
                            rlr_normalize(q.data(), dim);
                            const int32_t st = rlr_search_topk(ix, q.data(), 1, 50, -1.0f, lrows.data(), cosv.data(), nout.data());
                            if (st != RLR_OK && !g_said[0].exchange(true))
                                fprintf(stderr, "(rlr_search_topk on the stub: %d %s)\n", st, rlr_last_error());
                            break;
                        }
                        default: {
                            for (uint32_t i = 0; i < 64; ++i) {
                                prow[i] = next(&s) % n_rows;
                                psc[i] = 1.0f - 0.01f * static_cast<float>(i);
                            }
                            uint32_t n_sel = 0;
                            const int32_t st = rlr_mmr_select(ix, prow.data(), psc.data(), 64, 20, 0.3f, order.data(), mmr.data(), &n_sel);
                            if (st != RLR_OK && !g_said[1].exchange(true))
                                fprintf(stderr, "(rlr_mmr_select on the stub: %d %s)\n", st, rlr_last_error());
                            break;
                        }
                        }
                    }
                });
            for (auto &t : ts)
                t.join();
            printf("round %d, %2d threads done\n", rd, nt);
            fflush(stdout);
        }
        // mutation between rounds, alone (the engine's write lock): grow both indexes past the workspaces' accumulators,
        // then drop a few rows from both
        const uint64_t add = 3000;
        std::vector<float> more(add * dim, 0.25f);
        uint64_t first = 0;
        CHECK(rlr_index_append(ix, more.data(), add, 1, &first));
        for (uint64_t r = 0; r < add; ++r) {
            const std::string t = words(&g_seed, 20);
            CHECK(rlr_lexical_add_chunk(lx, first + r, t.data(), t.size()));
        }
        n_rows += add;
        std::vector<uint64_t> drop = {5, 17, n_rows - 1};
        CHECK(rlr_index_delete_rows(ix, drop.data(), static_cast<uint32_t>(drop.size())));
        CHECK(rlr_lexical_remove_rows(lx, drop.data(), static_cast<uint32_t>(drop.size())));
        n_rows -= drop.size();
    }
    // the sharded corpus: four shards on the one (stub) device, persistent shard workers, more callers than exchange
    // workspaces; the cross-shard MMR's gather / device-to-device copies / staged MMR run from several host threads at once
    {
        rlr_multi *m = nullptr;
        const int32_t devs[4] = {0, 0, 0, 0};
        CHECK(rlr_multi_create(dim, RLR_F32, 4, devs, &m));
        CHECK(rlr_multi_upload(m, rows.data(), 6000, 1));
        std::vector<std::thread> ts;
        for (int t = 0; t < 8; ++t)
            ts.emplace_back([&, t] {
                uint64_t s = 99 + t;
                std::vector<float> q(dim);
                std::vector<rlr_search_hit> hits(400);
                std::vector<uint64_t> prow(4 * 32);
                std::vector<float> psc(4 * 32), mmr(4 * 32);
                std::vector<uint32_t> order(4 * 32), nsel(4), sizes(4, 32);
                for (int it = 0; it < iters / 2; ++it) {
                    for (auto &v : q)
                        v = static_cast<float>(static_cast<int64_t>(next(&s) % 2001) - 1000) / 1000.0f;
                    uint32_t n = 0;
                    if ((t + it) % 2) {
                        // (kernels do nothing on the stub, so the rows a search returns are not valid pool rows: the engine call may
                        // refuse them -- what is under test is the shard workers and the merge, not the answer)
                        (void)rlr_multi_engine_search_with_diversity(m, q.data(), dim, 10, 0.5f, nullptr, nullptr, nullptr, 0, hits.data(), 400, &n);
                    } else {
                        for (size_t i = 0; i < prow.size(); ++i) {
                            prow[i] = next(&s) % 6000;
                            psc[i] = 1.0f - 0.01f * static_cast<float>(i % 32);
                        }
                        CHECK(rlr_multi_mmr_select_batch(m, prow.data(), psc.data(), sizes.data(), 4, 32, 8, 0.4f, order.data(), mmr.data(), nsel.data()));
                    }
                }
            });
        for (auto &t : ts)
            t.join();
        printf("multi engine, 8 threads done\n");
        CHECK(rlr_multi_destroy(m));
    }
    rlr_lexical_destroy(lx);
    CHECK(rlr_index_destroy(ix));
    if (g_fail.load()) {
        fprintf(stderr, "tsan_stress: %d calls failed\n", g_fail.load());
        return 1;
    }
    printf("tsan_stress ok\n");
    return 0;
}
