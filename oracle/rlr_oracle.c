/*
 * rlr_oracle.c -- CPU restatement of rust-local-rag's search_documents hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the *checker*: it may be imported,
 * linked or executed only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  The product path (rust-local-rag_amd/) never calls it and
 * must fail loudly when the HIP library is missing.
 *
 * Parity status: PINNED by the reference's own literal test vectors
 * (tests/golden/reference_kats.json, transcribed from
 * /root/reference/src/rag_engine.rs:2674-3226); the reference itself (Rust,
 * no rustc in this image) cannot be compiled here.  Everything above those
 * KATs (search-level ordering at N>3) is pinned only by this restatement --
 * see DESIGN.md "Oracle".
 *
 * Arithmetic contract: IEEE-754 binary32, round to
 * nearest even, strict left-to-right accumulation, products rounded before
 * they are added (rustc never contracts a*b+c and never reassociates float
 * reductions).  Build with:  gcc -O2 -ffp-contract=off  (NO -ffast-math).
 * The volatile-free loops below rely on those flags; oracle/Makefile sets them.
 *
 * Each function cites the reference lines it follows (paths relative to
 * /root/reference).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__FAST_MATH__)
#error "the oracle must not be built with -ffast-math"
#endif

#define RLR_O_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* a1  normalize            src/rag_engine.rs:1763-1771                      */
/* ------------------------------------------------------------------------ */
RLR_O_API void rlr_o_normalize(float *v, size_t n)
{
    float norm_sq = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float p = v[i] * v[i];
        norm_sq = norm_sq + p;
    }
    if (norm_sq > 1e-20f) {
        float norm = sqrtf(norm_sq);
        for (size_t i = 0; i < n; ++i)
            v[i] = v[i] / norm;
    }
}

/* ------------------------------------------------------------------------ */
/* a2  dot_product          src/rag_engine.rs:1776-1779 (zip truncates)      */
/* ------------------------------------------------------------------------ */
RLR_O_API float rlr_o_dot(const float *a, size_t na, const float *b, size_t nb)
{
    size_t n = na < nb ? na : nb;
    float s = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float p = a[i] * b[i];
        s = s + p;
    }
    return s;
}

/* ------------------------------------------------------------------------ */
/* a3  cosine_similarity    src/rag_engine.rs:1741-1759                      */
/* ------------------------------------------------------------------------ */
RLR_O_API float rlr_o_cosine(const float *a, size_t na, const float *b, size_t nb)
{
    if (na != nb)
        return 0.0f;
    const float eps = 1e-10f;
    float d = 0.0f, sa = 0.0f, sb = 0.0f;
    for (size_t i = 0; i < na; ++i) {
        float p = a[i] * b[i];
        d = d + p;
    }
    for (size_t i = 0; i < na; ++i) {
        float p = a[i] * a[i];
        sa = sa + p;
    }
    for (size_t i = 0; i < nb; ++i) {
        float p = b[i] * b[i];
        sb = sb + p;
    }
    float norm_a = sqrtf(sa), norm_b = sqrtf(sb);
    if (norm_a < eps || norm_b < eps)
        return 0.0f;
    float den = norm_a * norm_b;
    float c = d / den;
    /* f32::clamp(-1, 1): NaN stays NaN */
    if (c < -1.0f) c = -1.0f;
    if (c > 1.0f) c = 1.0f;
    return c;
}

/* ------------------------------------------------------------------------ */
/* a10 resolve_weight       src/rag_engine.rs:1869-1873, defaults :1801-1804 */
/* ------------------------------------------------------------------------ */
RLR_O_API float rlr_o_resolve_weight(int has_override, float w, float dflt)
{
    if (has_override && isfinite(w) && w >= 0.0f && w <= 1.0f)
        return w;
    return dflt;
}

/* ------------------------------------------------------------------------ */
/* HOT LOOP 1: per-row dot  src/rag_engine.rs:524-541 (embedding_score only) */
/* rows: n x d row-major; q: dq floats (already normalised by the caller).   */
/* ------------------------------------------------------------------------ */
RLR_O_API void rlr_o_scan(const float *rows, size_t n, size_t d, const float *q,
                          size_t dq, float *e_out)
{
    for (size_t r = 0; r < n; ++r)
        e_out[r] = rlr_o_dot(q, dq, rows + r * d, d);
}

struct scan_job {
    const float *rows;
    size_t r0, r1, d;
    const float *q;
    size_t dq;
    float *e_out;
};

static void *scan_worker(void *p)
{
    struct scan_job *j = (struct scan_job *)p;
    for (size_t r = j->r0; r < j->r1; ++r)
        j->e_out[r] = rlr_o_dot(j->q, j->dq, j->rows + r * j->d, j->d);
    return NULL;
}

/* "reference arithmetic, parallelised -- not something the reference does"
 * (BASELINE.md section 3): rows split over threads, per-row order unchanged. */
RLR_O_API void rlr_o_scan_mt(const float *rows, size_t n, size_t d, const float *q,
                             size_t dq, float *e_out, int n_threads)
{
    if (n_threads <= 1 || n < (size_t)n_threads) {
        rlr_o_scan(rows, n, d, q, dq, e_out);
        return;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    struct scan_job *jobs = (struct scan_job *)malloc(sizeof(struct scan_job) * (size_t)n_threads);
    size_t per = (n + (size_t)n_threads - 1) / (size_t)n_threads;
    for (int t = 0; t < n_threads; ++t) {
        size_t r0 = per * (size_t)t, r1 = r0 + per;
        if (r0 > n) r0 = n;
        if (r1 > n) r1 = n;
        jobs[t] = (struct scan_job){rows, r0, r1, d, q, dq, e_out};
        pthread_create(&th[t], NULL, scan_worker, &jobs[t]);
    }
    for (int t = 0; t < n_threads; ++t)
        pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

/* ------------------------------------------------------------------------ */
/* Ordering.  The reference sorts with                                        */
/*   scores.sort_by(|a,b| b.0.partial_cmp(&a.0).unwrap_or(Equal))  (:543,:669)*/
/* a *stable* merge sort over candidates visited in HashSet/HashMap order     */
/* (random per process), so exact ties -- and any NaN -- are resolved         */
/* arbitrarily upstream.  This build DEFINES the unspecified part:            */
/*   visiting order = ascending row index  => ties keep the lower row first;  */
/*   NaN scores order after every number (and keep row order among NaNs).     */
/* ------------------------------------------------------------------------ */
struct scored {
    float c; /* combined */
    float e; /* embedding */
    float l; /* lexical (normalised) */
    uint64_t row;
};

/* returns 1 when a must come strictly before b */
static int scored_before(const struct scored *a, const struct scored *b)
{
    int an = isnan(a->c), bn = isnan(b->c);
    if (an || bn)
        return !an && bn;
    return a->c > b->c;
}

static void merge_sort_scored(struct scored *v, struct scored *tmp, size_t n)
{
    if (n < 2)
        return;
    size_t h = n / 2;
    merge_sort_scored(v, tmp, h);
    merge_sort_scored(v + h, tmp, n - h);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* stable: take from the right run only when it is strictly before */
        if (scored_before(&v[j], &v[i]))
            tmp[k++] = v[j++];
        else
            tmp[k++] = v[i++];
    }
    while (i < h) tmp[k++] = v[i++];
    while (j < n) tmp[k++] = v[j++];
    memcpy(v, tmp, n * sizeof(*v));
}

/* ------------------------------------------------------------------------ */
/* a5+a6  RagEngine::search, exact-scan branch (ann_index == None, :502),     */
/*        reranker absent (:593-596 -> fallback :667-698).                    */
/*        src/rag_engine.rs:470-565, :599-700                                 */
/*                                                                            */
/* q_raw   : raw query embedding (dq floats); normalised here (:494) when     */
/*           normalize_query != 0.                                            */
/* lex_*   : LexicalIndex::score output restricted to rows of this corpus     */
/*           ({row: bm25 > 0}), n_lex <= 5*top_k (:505-506). May be empty.    */
/* stage   : 0 -> final results (first top_k of the candidates, :667-698)     */
/*           1 -> the initial_k candidate list handed to a reranker (:544-561)*/
/* outputs : rows / combined / embedding / lexical, at most `cap` entries.    */
/* returns : number of results.                                               */
/* ------------------------------------------------------------------------ */
RLR_O_API size_t rlr_o_search(const float *rows, size_t n, size_t d, const float *q_raw,
                              size_t dq, size_t top_k, float w_e, float w_l,
                              const uint64_t *lex_rows, const float *lex_scores, size_t n_lex,
                              int normalize_query, int stage, uint64_t *out_rows,
                              float *out_c, float *out_e, float *out_l, size_t cap)
{
    if (n == 0) /* :476-478 */
        return 0;
    if (top_k < 1) /* :490 */
        top_k = 1;

    float *q = (float *)malloc(sizeof(float) * (dq ? dq : 1));
    memcpy(q, q_raw, sizeof(float) * dq);
    if (normalize_query)
        rlr_o_normalize(q, dq); /* :494 */

    /* :515-519  max_lexical = fold(0, max).max(EPSILON) */
    float max_lex = 0.0f;
    for (size_t i = 0; i < n_lex; ++i)
        max_lex = fmaxf(max_lex, lex_scores[i]); /* f32::max ignores NaN like fmaxf */
    if (!(max_lex >= 1.1920929e-07f))
        max_lex = 1.1920929e-07f;

    /* sparse lexical term, densified for the oracle (tests are small) */
    float *lex = (float *)calloc(n, sizeof(float));
    unsigned char *has_lex = (unsigned char *)calloc(n, 1);
    for (size_t i = 0; i < n_lex; ++i) {
        if (lex_rows[i] < n) {
            lex[lex_rows[i]] = lex_scores[i];
            has_lex[lex_rows[i]] = 1;
        }
    }

    struct scored *sc = (struct scored *)malloc(sizeof(struct scored) * n);
    struct scored *tmp = (struct scored *)malloc(sizeof(struct scored) * n);
    for (size_t r = 0; r < n; ++r) { /* :524-541 */
        float e = rlr_o_dot(q, dq, rows + r * d, d);
        float l = has_lex[r] ? lex[r] / max_lex : 0.0f;
        float t0 = w_e * e;
        float t1 = w_l * l;
        float c = t0 + t1;
        sc[r] = (struct scored){c, e, l, (uint64_t)r};
    }
    merge_sort_scored(sc, tmp, n); /* :543 */

    size_t want = top_k * 3 > top_k ? top_k * 3 : top_k; /* :544 */
    size_t initial_k = n < want ? n : want;
    size_t n_out = stage ? initial_k : (initial_k < top_k ? initial_k : top_k); /* :667-698 */
    if (n_out > cap)
        n_out = cap;
    for (size_t i = 0; i < n_out; ++i) {
        out_rows[i] = sc[i].row;
        if (out_c) out_c[i] = sc[i].c;
        if (out_e) out_e[i] = sc[i].e;
        if (out_l) out_l[i] = sc[i].l;
    }
    free(sc);
    free(tmp);
    free(lex);
    free(has_lex);
    free(q);
    return n_out;
}

/* ------------------------------------------------------------------------ */
/* a11 get_embedding_candidates   src/rag_engine.rs:415-461 (None arm)        */
/* ------------------------------------------------------------------------ */
RLR_O_API size_t rlr_o_embedding_candidates(const float *rows, size_t n, size_t d,
                                            const float *q_raw, size_t dq, size_t count,
                                            uint64_t *out_rows, float *out_e)
{
    if (n == 0)
        return 0;
    float *q = (float *)malloc(sizeof(float) * (dq ? dq : 1));
    memcpy(q, q_raw, sizeof(float) * dq);
    rlr_o_normalize(q, dq);
    struct scored *sc = (struct scored *)malloc(sizeof(struct scored) * n);
    struct scored *tmp = (struct scored *)malloc(sizeof(struct scored) * n);
    for (size_t r = 0; r < n; ++r) {
        float e = rlr_o_dot(q, dq, rows + r * d, d);
        sc[r] = (struct scored){e, e, 0.0f, (uint64_t)r};
    }
    merge_sort_scored(sc, tmp, n);
    size_t n_out = n < count ? n : count;
    for (size_t i = 0; i < n_out; ++i) {
        out_rows[i] = sc[i].row;
        out_e[i] = sc[i].e;
    }
    free(sc);
    free(tmp);
    free(q);
    return n_out;
}

/* ------------------------------------------------------------------------ */
/* f2  reranker blend + result assembly   src/rag_engine.rs:599-700           */
/* cand_*: the initial_k stage-1 candidates in the order `search` produced    */
/*         them (row, initial_score).  rer_*: the reranker's output in its    */
/*         order (row, relevance); rows not among the candidates or repeated  */
/*         are skipped (:617-618).  Output: final order, score, and whether a */
/*         reranker score is attached (0 for fallback-filled results).        */
/* Tie rule for the fallback sort over `candidate_map.values()` (HashMap      */
/* order in the reference): candidate order.                                  */
/* ------------------------------------------------------------------------ */
struct blended {
    float score;
    size_t cand;  /* index into the candidate list */
    float rer;
    int has_rer;
};

static void stable_sort_blended(struct blended *v, struct blended *tmp, size_t n)
{
    if (n < 2)
        return;
    size_t h = n / 2;
    stable_sort_blended(v, tmp, h);
    stable_sort_blended(v + h, tmp, n - h);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* b.score.partial_cmp(&a.score).unwrap_or(Equal): NaN ties with everything */
        int right_first = (v[j].score > v[i].score) && !isnan(v[j].score) && !isnan(v[i].score);
        if (right_first)
            tmp[k++] = v[j++];
        else
            tmp[k++] = v[i++];
    }
    while (i < h) tmp[k++] = v[i++];
    while (j < n) tmp[k++] = v[j++];
    memcpy(v, tmp, n * sizeof(*v));
}

RLR_O_API size_t rlr_o_blend(const uint64_t *cand_rows, const float *cand_initial, size_t n_cand,
                             const uint64_t *rer_rows, const float *rer_relevance, size_t n_rer,
                             size_t top_k, float w_reranker, float w_initial, uint32_t *out_cand,
                             float *out_score, float *out_rer, int32_t *out_has_rer)
{
    struct blended *res = (struct blended *)malloc(sizeof(struct blended) * (n_cand + n_rer + 1));
    struct blended *tmp = (struct blended *)malloc(sizeof(struct blended) * (n_cand + n_rer + 1));
    unsigned char *seen = (unsigned char *)calloc(n_cand + 1, 1);
    size_t n_res = 0;
    if (n_rer > 0) { /* :602 */
        float max_rer = 0.0f, max_init = 0.0f;
        for (size_t i = 0; i < n_rer; ++i) max_rer = fmaxf(max_rer, rer_relevance[i]);
        if (!(max_rer >= 1.1920929e-07f)) max_rer = 1.1920929e-07f;
        for (size_t i = 0; i < n_cand; ++i) max_init = fmaxf(max_init, cand_initial[i]);
        if (!(max_init >= 1.1920929e-07f)) max_init = 1.1920929e-07f;
        for (size_t i = 0; i < n_rer; ++i) { /* :616-654 */
            size_t c = n_cand;
            for (size_t j = 0; j < n_cand; ++j)
                if (cand_rows[j] == rer_rows[i]) { c = j; break; }
            if (c == n_cand || seen[c])
                continue;
            seen[c] = 1;
            float rn = rer_relevance[i] / max_rer;
            float in = cand_initial[c] / max_init;
            float t0 = w_reranker * rn;
            float t1 = w_initial * in;
            res[n_res++] = (struct blended){t0 + t1, c, rer_relevance[i], 1};
        }
        stable_sort_blended(res, tmp, n_res); /* :657-661 */
        if (n_res > top_k) n_res = top_k;     /* :664 */
    }
    if (n_res < top_k) { /* :667-698 fallback fill by initial score */
        struct blended *fb = (struct blended *)malloc(sizeof(struct blended) * (n_cand + 1));
        for (size_t j = 0; j < n_cand; ++j)
            fb[j] = (struct blended){cand_initial[j], j, 0.0f, 0};
        stable_sort_blended(fb, tmp, n_cand);
        for (size_t j = 0; j < n_cand && n_res < top_k; ++j)
            if (!seen[fb[j].cand]) {
                seen[fb[j].cand] = 1;
                res[n_res++] = fb[j];
            }
        free(fb);
    }
    for (size_t i = 0; i < n_res; ++i) {
        out_cand[i] = (uint32_t)res[i].cand;
        out_score[i] = res[i].score;
        out_rer[i] = res[i].rer;
        out_has_rer[i] = res[i].has_rer;
    }
    free(res);
    free(tmp);
    free(seen);
    return n_res;
}

/* ------------------------------------------------------------------------ */
/* a8  mmr_diversify        src/rag_engine.rs:767-839 (test twin :2824-2875)  */
/* emb: P x d row-major candidate embeddings in candidate order;              */
/* score: P relevance scores (the combined score of each result).             */
/* order_out: indices into the candidate list, in pick order (<= P entries).  */
/* mmr_out (optional): the MMR value of each pick (the reference only logs it;*/
/*          first pick has none -> NaN).                                      */
/* ------------------------------------------------------------------------ */
RLR_O_API size_t rlr_o_mmr(const float *emb, const float *score, size_t P, size_t d,
                           size_t top_k, float lambda, uint32_t *order_out, float *mmr_out)
{
    if (P == 0) /* :773-775 */
        return 0;
    uint32_t *rem = (uint32_t *)malloc(sizeof(uint32_t) * P);
    size_t n_rem = P, n_sel = 0;
    for (size_t i = 0; i < P; ++i)
        rem[i] = (uint32_t)i;

    /* :782-785  first = remaining.swap_remove(0) -- unconditional, so top_k == 0 still yields 1 */
    order_out[n_sel] = rem[0];
    if (mmr_out) mmr_out[n_sel] = NAN;
    n_sel++;
    rem[0] = rem[n_rem - 1];
    n_rem--;

    const float one_minus = 1.0f - lambda;
    while (n_sel < top_k && n_rem > 0) { /* :788 */
        float best = -INFINITY;
        size_t best_idx = 0;
        for (size_t idx = 0; idx < n_rem; ++idx) { /* :792 */
            uint32_t c = rem[idx];
            float rel = score[c];
            if (!isfinite(rel)) /* :794-797 */
                continue;
            float ms = 0.0f; /* :800-804 fold(0.0, max) over finite sims */
            for (size_t s = 0; s < n_sel; ++s) {
                float sim = rlr_o_dot(emb + (size_t)c * d, d, emb + (size_t)order_out[s] * d, d);
                if (isfinite(sim))
                    ms = fmaxf(ms, sim);
            }
            float t0 = one_minus * rel; /* :808-809 */
            float t1 = lambda * ms;
            float m = t0 - t1;
            if (isfinite(m) && m > best) { /* :812-815 strict > */
                best = m;
                best_idx = idx;
            }
        }
        if (best == -INFINITY) /* :819-822 */
            break;
        order_out[n_sel] = rem[best_idx]; /* :825 swap_remove(best_idx) */
        if (mmr_out) mmr_out[n_sel] = best;
        n_sel++;
        rem[best_idx] = rem[n_rem - 1];
        n_rem--;
    }
    free(rem);
    return n_sel;
}

/* ------------------------------------------------------------------------ */
/* a7  search_with_diversity   src/rag_engine.rs:717-759                      */
/* API-layer defaults/caps (mcp_server.rs:85-86, :364, :375-376) are applied  */
/* by the caller; this function clamps lambda like :725.                      */
/* ------------------------------------------------------------------------ */
RLR_O_API size_t rlr_o_search_with_diversity(const float *rows, size_t n, size_t d,
                                             const float *q_raw, size_t dq, size_t top_k,
                                             float diversity, float w_e, float w_l,
                                             const uint64_t *lex_rows, const float *lex_scores,
                                             size_t n_lex, int normalize_query,
                                             uint64_t *out_rows, float *out_c, float *out_e,
                                             float *out_l, size_t cap)
{
    /* f32::clamp(0,1): NaN stays NaN, and NaN == 0.0 is false -> MMR branch */
    if (diversity < 0.0f) diversity = 0.0f;
    if (diversity > 1.0f) diversity = 1.0f;
    if (diversity == 0.0f) /* :728-730 */
        return rlr_o_search(rows, n, d, q_raw, dq, top_k, w_e, w_l, lex_rows, lex_scores,
                            n_lex, normalize_query, 0, out_rows, out_c, out_e, out_l, cap);

    size_t pool = top_k * 3 > top_k + 10 ? top_k * 3 : top_k + 10; /* :734 */
    uint64_t *p_rows = (uint64_t *)malloc(sizeof(uint64_t) * (pool ? pool : 1));
    float *p_c = (float *)malloc(sizeof(float) * (pool ? pool : 1));
    float *p_e = (float *)malloc(sizeof(float) * (pool ? pool : 1));
    float *p_l = (float *)malloc(sizeof(float) * (pool ? pool : 1));
    size_t P = rlr_o_search(rows, n, d, q_raw, dq, pool, w_e, w_l, lex_rows, lex_scores, n_lex,
                            normalize_query, 0, p_rows, p_c, p_e, p_l, pool); /* :735 */
    size_t n_out = 0;
    if (P > 0) {
        float *emb = (float *)malloc(sizeof(float) * P * d); /* :742-753 */
        for (size_t i = 0; i < P; ++i)
            memcpy(emb + i * d, rows + p_rows[i] * d, sizeof(float) * d);
        uint32_t *order = (uint32_t *)malloc(sizeof(uint32_t) * P);
        size_t k = rlr_o_mmr(emb, p_c, P, d, top_k, diversity, order, NULL); /* :756 */
        n_out = k < cap ? k : cap;
        for (size_t i = 0; i < n_out; ++i) {
            out_rows[i] = p_rows[order[i]];
            if (out_c) out_c[i] = p_c[order[i]];
            if (out_e) out_e[i] = p_e[order[i]];
            if (out_l) out_l[i] = p_l[order[i]];
        }
        free(order);
        free(emb);
    }
    free(p_rows);
    free(p_c);
    free(p_e);
    free(p_l);
    return n_out;
}

/* ------------------------------------------------------------------------ */
/* Synthetic corpus generator (NOT in the reference).         */
/* Integer-only so that host and device produce bit-identical rows:           */
/*   h   = splitmix64 finaliser of a per-element counter                      */
/*   raw = (sum of the four 16-bit fields of h) - 131070     (Irwin-Hall n=4) */
/*   optional cluster centre added (2x weight) to create near-duplicates;     */
/*   bit 31 of n_clusters ("tight"): the per-row part is divided by 16 first  */
/*   (arithmetic shift), so the rows of a cluster are near-copies of each     */
/*   other (cosine ~0.999): a dense top-of-the-ranking, as boilerplate chunks */
/*   produce it                                                               */
/*   row = reference normalize() of the raw row.                              */
/* The HIP twin is rust-local-rag_amd/csrc/synth.hip; tests compare the two.  */
/* ------------------------------------------------------------------------ */
static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline int32_t ih4(uint64_t h)
{
    return (int32_t)((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48)) - 131070;
}

RLR_O_API float rlr_o_synth_raw(uint64_t seed, uint64_t row, uint32_t col, uint32_t d,
                                uint32_t n_clusters)
{
    uint64_t s = mix64(seed ^ 0x5EED5EED5EED5EEDULL);
    uint64_t idx = row * (uint64_t)d + col;
    int32_t t = ih4(mix64(s + (idx + 1) * 0x9E3779B97F4A7C15ULL));
    const int tight = (n_clusters & 0x80000000u) != 0;
    n_clusters &= 0x7FFFFFFFu;
    if (n_clusters) {
        uint64_t cl = mix64(s ^ (row + 0x632BE59BD9B4E019ULL)) % n_clusters;
        uint64_t cidx = cl * (uint64_t)d + col;
        int32_t c = ih4(mix64((s ^ 0xC1A57E55C1A57E55ULL) + (cidx + 1) * 0x9E3779B97F4A7C15ULL));
        if (tight)
            t = t / 16; /* toward zero, as the device twin */
        t += 2 * c;
    }
    return (float)t * (1.0f / 65536.0f); /* |t| < 2^24: exact */
}

RLR_O_API void rlr_o_synth_rows(float *out, uint64_t row0, size_t n, uint32_t d, uint64_t seed,
                                uint32_t n_clusters)
{
    for (size_t r = 0; r < n; ++r) {
        float *v = out + r * d;
        for (uint32_t c = 0; c < d; ++c)
            v[c] = rlr_o_synth_raw(seed, row0 + r, c, d, n_clusters);
        rlr_o_normalize(v, d);
    }
}

/* ------------------------------------------------------------------------ */
/* binary16 helpers for the fp16-storage configs (C5): rows are rounded to    */
/* fp16 AFTER normalisation and the f32 reference arithmetic runs on the      */
/* exactly-widened values (note under the config table).         */
/* ------------------------------------------------------------------------ */
RLR_O_API uint16_t rlr_o_f32_to_f16(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? (0x200u | ((ax >> 13) & 0x3FFu)) : 0));
    if (ax >= 0x477FF000u) /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7C00u);
    if (ax < 0x33000001u) /* <= 2^-25 rounds to zero (ties-to-even at exactly 2^-25) */
        return (uint16_t)sign;
    int32_t exp = (int32_t)(ax >> 23) - 127;
    uint32_t man = (ax & 0x7FFFFFu) | 0x800000u;
    uint32_t half;
    if (exp < -14) { /* subnormal half */
        int shift = 13 + (-14 - exp);
        uint32_t q = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (q & 1u)))
            q++;
        half = q;
    } else {
        uint32_t q = ((uint32_t)(exp + 15) << 10) | ((man >> 13) & 0x3FFu);
        uint32_t rem = man & 0x1FFFu;
        if (rem > 0x1000u || (rem == 0x1000u && (q & 1u)))
            q++;
        half = q;
    }
    return (uint16_t)(sign | half);
}

RLR_O_API float rlr_o_f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t x;
    if (exp == 0) {
        if (man == 0) {
            x = sign;
        } else {
            int e = -1;
            do {
                e++;
                man <<= 1;
            } while (!(man & 0x400u));
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) {
        x = sign | 0x7F800000u | (man << 13);
    } else {
        x = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

RLR_O_API void rlr_o_round_rows_f16(float *rows, size_t count)
{
    for (size_t i = 0; i < count; ++i)
        rows[i] = rlr_o_f16_to_f32(rlr_o_f32_to_f16(rows[i]));
}

RLR_O_API const char *rlr_o_build_flags(void)
{
#if defined(__FP_FAST_FMAF)
    return "WARNING: fused multiply-add contraction may be enabled";
#else
    return "gcc -O2 -ffp-contract=off (no fast-math)";
#endif
}
