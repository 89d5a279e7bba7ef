/* c_step.c -- the per-query cost of rlr_search_topk from plain C (no Python between two steps): rows x 768 f32, top-100.
 *   gcc -O2 -I include scratch/c_step.c -L rust-local-rag_amd -lrlr_gpu -lm -Wl,-rpath,$PWD/rust-local-rag_amd -o /tmp/c_step
 *   /tmp/c_step [rows] [steps] */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "rlr_gpu.h"

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}

int main(int argc, char **argv)
{
    const uint64_t n = argc > 1 ? strtoull(argv[1], NULL, 10) : 1250000;
    const int steps = argc > 2 ? atoi(argv[2]) : 300;
    const uint32_t dim = 768, k = 100;
    rlr_index *ix = NULL;
    if (rlr_index_create(dim, RLR_F32, 0, &ix) != RLR_OK || rlr_index_fill_synthetic(ix, n, 0, 0x5EED0003ull, 0) != RLR_OK) {
        fprintf(stderr, "setup failed: %s\n", rlr_last_error());
        return 1;
    }
    float *q = (float *)malloc(sizeof(float) * dim * 64);
    for (int j = 0; j < 64; ++j) {
        double s2 = 0;
        for (uint32_t i = 0; i < dim; ++i) {
            q[j * dim + i] = sinf(0.37f * (float)i + (float)j) + 0.25f * cosf(0.11f * (float)(i + 3 * j));
            s2 += (double)q[j * dim + i] * q[j * dim + i];
        }
        for (uint32_t i = 0; i < dim; ++i)
            q[j * dim + i] = (float)(q[j * dim + i] / sqrt(s2));
    }
    uint64_t rows[100];
    float cos[100];
    uint32_t got = 0;
    for (int i = 0; i < 40; ++i)
        rlr_search_topk(ix, q + (i % 64) * dim, 1, k, -1.0f, rows, cos, &got);
    const double t0 = now();
    for (int i = 0; i < steps; ++i)
        if (rlr_search_topk(ix, q + (i % 64) * dim, 1, k, -1.0f, rows, cos, &got) != RLR_OK) {
            fprintf(stderr, "search failed: %s\n", rlr_last_error());
            return 1;
        }
    printf("C loop: %llu rows, %.1f us per rlr_search_topk (top row %llu, %u results)\n", (unsigned long long)n,
           (now() - t0) / steps * 1e6, (unsigned long long)rows[0], got);
    rlr_index_destroy(ix);
    return 0;
}
