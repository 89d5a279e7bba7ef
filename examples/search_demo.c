/* search_demo.c -- the C ABI from plain C, no Python, no torch: what a Rust/C host links.
 *
 *   gcc -O2 -I include examples/search_demo.c -L rust-local-rag_amd -lrlr_gpu -lm \
 *       -Wl,-rpath,$PWD/rust-local-rag_amd -o search_demo
 *   ./search_demo [rows] [dim] [k]
 *
 * Fills an index with the deterministic synthetic corpus, runs one top-k search and one MMR
 * selection through rlr_engine_*, then the same with a query TEXT against a small GPU BM25 index
 * (rlr_lexical_* + rlr_engine_search_text: the hybrid search of search_documents), and prints the
 * results as "row score" lines (the GPU test compares them with the Python binding's results for
 * the same inputs).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rlr_engine.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        int32_t s_ = (call);                                                           \
        if (s_ != RLR_OK) {                                                            \
            fprintf(stderr, "%s -> %d: %s\n", #call, s_, rlr_last_error());           \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

int main(int argc, char **argv)
{
    const uint64_t n = argc > 1 ? strtoull(argv[1], NULL, 10) : 100000;
    const uint32_t dim = argc > 2 ? (uint32_t)atoi(argv[2]) : 768;
    const uint32_t k = argc > 3 ? (uint32_t)atoi(argv[3]) : 5;
    if (rlr_device_count() < 1) {
        fprintf(stderr, "no GPU: librlr_gpu has no CPU path\n");
        return 2;
    }
    rlr_index *ix = NULL;
    CHECK(rlr_index_create(dim, RLR_F32, 0, &ix));
    CHECK(rlr_index_fill_synthetic(ix, n, 0, 0x5EED0003ull, 32));

    /* a raw (un-normalised) query: simple deterministic ramp; the engine normalises it */
    float *q = (float *)malloc(sizeof(float) * dim);
    for (uint32_t i = 0; i < dim; ++i)
        q[i] = sinf(0.37f * (float)i) + 0.25f * cosf(0.11f * (float)i);

    rlr_search_hit *hits = (rlr_search_hit *)malloc(sizeof(rlr_search_hit) * (3 * k + 10));
    uint32_t n_hits = 0;
    CHECK(rlr_engine_search(ix, q, dim, k, NULL, NULL, NULL, 0, 0, hits, 3 * k + 10, &n_hits));
    printf("search top_k=%u -> %u hits\n", k, n_hits);
    for (uint32_t i = 0; i < n_hits; ++i)
        printf("S %llu %.9g %.9g\n", (unsigned long long)hits[i].row, hits[i].score, hits[i].embedding_score);

    CHECK(rlr_engine_search_with_diversity(ix, q, dim, k, 0.3f, NULL, NULL, NULL, 0, hits, 3 * k + 10, &n_hits));
    printf("search_with_diversity lambda=0.3 -> %u hits\n", n_hits);
    for (uint32_t i = 0; i < n_hits; ++i)
        printf("D %llu %.9g %.9g\n", (unsigned long long)hits[i].row, hits[i].score, hits[i].embedding_score);

    /* hybrid: every 3rd of the first 3000 chunks carries words; the query text names two of them */
    rlr_lexical *lex = NULL;
    CHECK(rlr_lexical_create(0, &lex));
    char text[128], toks[128];
    for (uint64_t r = 0; r < 3000 && r < n; r += 3) {
        snprintf(text, sizeof text, "Chunk number %llu, about topic%llu and Theme%llu!", (unsigned long long)r,
                 (unsigned long long)(r % 7), (unsigned long long)(r % 11));
        size_t len = 0;
        CHECK(rlr_tokenize_ascii(text, strlen(text), toks, sizeof toks, &len)); /* lower-case, >= 3 bytes, split */
        CHECK(rlr_lexical_add_chunk(lex, r, toks, len));
    }
    const char *query_tokens = "topic3 theme5";
    CHECK(rlr_engine_search_text(ix, lex, q, dim, query_tokens, strlen(query_tokens), k, 0.3f, 0, NULL, hits, 3 * k + 10,
                                 &n_hits));
    printf("search_with_diversity + query text -> %u hits\n", n_hits);
    for (uint32_t i = 0; i < n_hits; ++i)
        printf("T %llu %.9g %.9g %.9g\n", (unsigned long long)hits[i].row, hits[i].score, hits[i].embedding_score,
               hits[i].lexical_score);
    rlr_lexical_destroy(lex);

    free(hits);
    free(q);
    CHECK(rlr_index_destroy(ix));
    return 0;
}
