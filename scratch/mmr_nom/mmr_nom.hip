// mmr_nom.hip -- mmr_diversify (reference src/rag_engine.rs:767-839) for BATCHES of pools with the similarity matrix
// NOMINATED on the matrix cores and every greedy step CERTIFIED (hard part 2).
//
// The reference recomputes dot(candidate, selected) for every remaining candidate in every round; the exact-order
// restatement of that (gram_tiled_kernel, exact.hip) is P(P+1)/2 strict left-to-right f32 chains per pool on the vector
// ALUs -- 3.3 ms per 1024 pools of 308 x 1024-d, most of config 5's MMR time.  Here:
//   1. gram_nom_kernel: the P x P matrix of every pool by v_mfma_f32_16x16x32_f16 over the binary16 rows (products of two
//      binary16 values are exact in f32; only the order and rounding of the f32 accumulation differ from the reference),
//      so |nominated - reference dot| <= eps = (3 dim + 64) 2^-24 |a| |b| (both accumulations, the matrix cores' at twice
//      the unit roundoff to allow for truncation);
//   2. mmr_greedy_cert_kernel: the register-resident greedy loop of exact.hip on the nominated values.  A step whose winner
//      leads every other candidate's MMR value by more than 2 (lambda eps + arithmetic slack) is CERTIFIED: the reference
//      picks the same candidate.  Otherwise the candidates within that margin are re-evaluated with REFERENCE-ORDER dots:
//      a candidate's exact max-similarity is attained among the selected rows whose nominated similarity lies within
//      2 eps of its nominated maximum (usually one or two), so a handful of strict-order chains -- one per lane -- decide
//      the step exactly, with the reference's visiting-order tie rule;
//   3. the logged MMR values of the k winners (mmr_out, optional) come from the same near-maximum argument: one or two
//      exact chains per pick instead of t.
// Pools the bound cannot vouch for (a non-finite row: its Gram diagonal is not finite; an ambiguity among more than
// kMaxAmbiguous candidates, e.g. many duplicated chunks) are flagged and go through the exact kernels.  Picks and logged
// values are bit-identical to exact.hip's and the oracle's.
#include "common.h"
#include "kernels.h"
#include "../../include/rlr_gpu.h"

namespace rlr {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kT = 64;                 // tile side (rows x rows)
constexpr int kKC = 64;                // binary16 elements per staged K chunk
constexpr int kRowPitch = kKC * 2 + 16; // bytes per staged row: 16-byte fragment reads of 16 consecutive rows hit 16 different slots
constexpr uint32_t kMaxAmbiguous = 8;  // more candidates than this inside one step's margin: the pool takes the exact path

__device__ inline bool finite_f(float x)
{
    return (__builtin_bit_cast(uint32_t, x) & 0x7F800000u) != 0x7F800000u;
}

// ---------------------------------------------------------------------------------------------------------------------
// 1. nominated Gram matrices.  Workgroup = 4 waves = one 64 x 64 tile (ti <= tj) of one pool, wave (wy, wx) owns a 32 x 32
// quadrant = 2 x 2 MFMA tiles; both row blocks are staged through LDS per 64-wide K chunk (next chunk's global loads in
// flight behind the current chunk's MFMAs); the tile is written to G[i][j] and G[j][i].  All tiles of a pool run on ONE XCD
// (blockIdx -> XCD round robin), so the pool's rows are fetched from HBM once and re-read from that XCD's L2.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_nom_kernel(const unsigned char *__restrict__ rows, uint32_t pitch_bytes, uint32_t dim,
                                                       const uint32_t *__restrict__ list, const uint32_t *__restrict__ sizes,
                                                       uint32_t P, uint32_t n_pools, uint32_t tiles_per_pool,
                                                       float *__restrict__ gram)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_ab[2][2][kT * kRowPitch];
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t pool = (slot / tiles_per_pool) * 8 + xcd;
    if (pool >= n_pools)
        return;
    uint32_t ti = 0, rem = slot % tiles_per_pool; // linear tile id -> (ti <= tj): row ti holds nt - ti tiles
    const uint32_t nt = (P + kT - 1) / kT;
    while (rem >= nt - ti) {
        rem -= nt - ti;
        ++ti;
    }
    const uint32_t tj = ti + rem;
    const uint32_t size = sizes ? sizes[pool] : P;
    const uint32_t i0 = ti * kT, j0 = tj * kT;
    if (size == 0 || i0 >= size || j0 >= size)
        return;
    list += static_cast<size_t>(pool) * P;
    gram += static_cast<size_t>(pool) * P * P;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t wy = wave >> 1, wx = wave & 1;
    // this thread's two 16-byte units per operand and chunk: rows t/8 and 32 + t/8, segment t % 8
    const uint32_t lr = t >> 3, seg = t & 7;
    const unsigned char *ga[2], *gb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t ra = min(i0 + lr + 32 * h, size - 1), rb = min(j0 + lr + 32 * h, size - 1);
        ga[h] = rows + static_cast<size_t>(list[ra]) * pitch_bytes + seg * 16;
        gb[h] = rows + static_cast<size_t>(list[rb]) * pitch_bytes + seg * 16;
    }
    const uint32_t n_chunks = dim / kKC;
    half8 pa[2], pb[2];
    auto fetch = [&](uint32_t c) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            pa[h] = *reinterpret_cast<const half8 *>(ga[h] + static_cast<size_t>(c) * (kKC * 2));
            pb[h] = *reinterpret_cast<const half8 *>(gb[h] + static_cast<size_t>(c) * (kKC * 2));
        }
    };
    auto stash = [&](uint32_t buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<half8 *>(&s_ab[buf][0][(lr + 32 * h) * kRowPitch + seg * 16]) = pa[h];
            *reinterpret_cast<half8 *>(&s_ab[buf][1][(lr + 32 * h) * kRowPitch + seg * 16]) = pb[h];
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
            acc[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    fetch(0);
    stash(0);
    __syncthreads();
    for (uint32_t c = 0; c < n_chunks; ++c) {
        const uint32_t buf = c & 1;
        const bool more = c + 1 < n_chunks;
        if (more)
            fetch(c + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 fa[2], fb[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                fa[r] = *reinterpret_cast<const half8 *>(&s_ab[buf][0][(32 * wy + 16 * r + (lane & 15)) * kRowPitch + ks * 64 + (lane >> 4) * 16]);
                fb[r] = *reinterpret_cast<const half8 *>(&s_ab[buf][1][(32 * wx + 16 * r + (lane & 15)) * kRowPitch + ks * 64 + (lane >> 4) * 16]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        if (more)
            stash(buf ^ 1);
        __syncthreads();
    }
    // accumulator element e of lane l: row 4 (l >> 4) + e of the A block, column l & 15 of the B block
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint32_t j = j0 + 32 * wx + 16 * b + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t i = i0 + 32 * wy + 16 * a + 4 * (lane >> 4) + e;
                if (i < size && j < size) {
                    gram[static_cast<size_t>(i) * P + j] = acc[a][b][e];
                    gram[static_cast<size_t>(j) * P + i] = acc[a][b][e];
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// reference-order dot of two stored rows (dot_product, rag_engine.rs:1777-1779): strict left-to-right f32 accumulation,
// each product rounded before its add, no FMA (-ffp-contract=off).  One lane = one pair.
// ---------------------------------------------------------------------------------------------------------------------
template <bool F16>
__device__ inline float dot_ref_pair(const float4 *__restrict__ ra, const float4 *__restrict__ rb, uint32_t dim)
{
    float s = 0.0f;
    constexpr uint32_t EPU = F16 ? 8 : 4;
    const uint32_t full = dim / EPU;
    uint32_t u = 0;
    for (; u + 4 <= full; u += 4) {
        float4 x[4], y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[i] = ra[u + i];
            y[i] = rb[u + i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (F16) {
                const uint32_t wx[4] = {__builtin_bit_cast(uint32_t, x[i].x), __builtin_bit_cast(uint32_t, x[i].y),
                                        __builtin_bit_cast(uint32_t, x[i].z), __builtin_bit_cast(uint32_t, x[i].w)};
                const uint32_t wy[4] = {__builtin_bit_cast(uint32_t, y[i].x), __builtin_bit_cast(uint32_t, y[i].y),
                                        __builtin_bit_cast(uint32_t, y[i].z), __builtin_bit_cast(uint32_t, y[i].w)};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p0 = h2f(static_cast<uint16_t>(wx[j] & 0xFFFF)) * h2f(static_cast<uint16_t>(wy[j] & 0xFFFF));
                    s = s + p0;
                    const float p1 = h2f(static_cast<uint16_t>(wx[j] >> 16)) * h2f(static_cast<uint16_t>(wy[j] >> 16));
                    s = s + p1;
                }
            } else {
                float p;
                p = x[i].x * y[i].x; s = s + p;
                p = x[i].y * y[i].y; s = s + p;
                p = x[i].z * y[i].z; s = s + p;
                p = x[i].w * y[i].w; s = s + p;
            }
        }
    }
    if constexpr (F16) {
        const uint16_t *ha = reinterpret_cast<const uint16_t *>(ra), *hb = reinterpret_cast<const uint16_t *>(rb);
        for (uint32_t e = u * EPU; e < dim; ++e) {
            const float p = h2f(ha[e]) * h2f(hb[e]);
            s = s + p;
        }
    } else {
        const float *fa = reinterpret_cast<const float *>(ra), *fb = reinterpret_cast<const float *>(rb);
        for (uint32_t e = u * EPU; e < dim; ++e) {
            const float p = fa[e] * fb[e];
            s = s + p;
        }
    }
    return s;
}

// exact max_similarity of candidate `c` over the first n_sel picks (s_order), the reference's
// `fold(0.0, |m, d| if d.is_finite() { m.max(d) } else { m })` (:800-804), from reference-order dots of only those picks
// whose NOMINATED similarity is within 2 eps of the candidate's nominated maximum g_star (no other pick can hold the exact
// maximum).  Wave-uniform arguments; every lane returns the value.
template <bool F16>
__device__ inline float exact_max_sim(const float *__restrict__ gram_c, float g_star, float eps, const uint32_t *s_order,
                                      uint32_t n_sel, const unsigned char *__restrict__ rows, uint32_t pitch_bytes, uint32_t dim,
                                      const uint32_t *__restrict__ list, uint32_t c, uint32_t lane)
{
    if (g_star + eps < 0.0f) // every exact similarity is negative: the fold stays at +0.0
        return 0.0f;
    const float near = g_star - 2.0f * eps;
    const float4 *rc = reinterpret_cast<const float4 *>(rows + static_cast<size_t>(list[c]) * pitch_bytes);
    float ms = 0.0f;
    for (uint32_t base = 0; base < n_sel; base += 64) {
        const uint32_t jj = base + lane;
        if (jj < n_sel) {
            const uint32_t s = s_order[jj];
            if (gram_c[s] >= near) {
                const float d = dot_ref_pair<F16>(rc, reinterpret_cast<const float4 *>(rows + static_cast<size_t>(list[s]) * pitch_bytes), dim);
                ms = (finite_f(d) & (d > ms)) ? d : ms;
            }
        }
    }
    return wave_max_f32_no_nan(ms); // every lane's value is >= +0.0 and finite
}

// ---------------------------------------------------------------------------------------------------------------------
// 2. + 3. the greedy loop on nominated values with certified steps (one wavefront per pool, candidates in registers as in
// exact.hip's mmr_greedy_reg_kernel, whose pick arithmetic this repeats operation for operation) and, when the caller wants
// the logged values, their exact recomputation by all four waves afterwards.
//   status[pool]: 0 = picks (and values) are the reference's; 1 = the pool needs the exact kernels (nothing written).
// ---------------------------------------------------------------------------------------------------------------------
template <int J, bool F16>
__global__ __launch_bounds__(256) void mmr_greedy_cert_kernel(const float *__restrict__ gram, const float *__restrict__ scores,
                                                              uint32_t P, uint32_t k, float lambda, float eps_rel,
                                                              uint32_t *__restrict__ out_order, float *__restrict__ out_mmr,
                                                              uint32_t *__restrict__ out_n, const uint32_t *__restrict__ sizes,
                                                              const unsigned char *__restrict__ rows, uint32_t pitch_bytes,
                                                              uint32_t dim, const uint32_t *__restrict__ list,
                                                              uint32_t *__restrict__ status, int want_values,
                                                              uint32_t *__restrict__ counters)
{
    __shared__ uint32_t s_order[1024];
    __shared__ float s_eps;
    __shared__ uint32_t s_nsel;
    const uint32_t stride = P;
    gram += static_cast<size_t>(blockIdx.x) * stride * stride;
    scores += static_cast<size_t>(blockIdx.x) * stride;
    list += static_cast<size_t>(blockIdx.x) * stride;
    out_order += static_cast<size_t>(blockIdx.x) * stride;
    out_mmr += static_cast<size_t>(blockIdx.x) * stride;
    out_n += blockIdx.x;
    status += blockIdx.x;
    if (sizes)
        P = sizes[blockIdx.x];
    if (P == 0) {
        if (threadIdx.x == 0) {
            *out_n = 0;
            *status = 0;
        }
        return;
    }
    {   // sweep the matrix into this XCD's L2, one 128-byte line per thread and step (see mmr_greedy_reg_kernel)
        float warm = 0.0f;
        const uint32_t n = P * stride;
        for (uint32_t i = threadIdx.x * 32; i < n; i += 256 * 32)
            warm += gram[i];
        asm volatile("" ::"v"(warm));
    }
    const uint32_t lane = threadIdx.x & 63;
    const float neg_inf = -__builtin_inff();
    const float nan_f = __builtin_bit_cast(float, 0x7FC00000u);
    const float one_minus = 1.0f - lambda;
    uint32_t n_sel = 1;
    if (threadIdx.x < 64) {
        float t0[J], ms[J], gmax[J];
        uint32_t pos[J], idx[J];
        float diag_max = 0.0f, t_abs = 0.0f;
        bool bad = false;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const uint32_t c = lane + 64 * j;
            idx[j] = min(c, P - 1);
            const float r = scores[idx[j]];
            const bool usable = (c < P) & (c != 0);
            t0[j] = usable ? one_minus * r : nan_f;
            ms[j] = 0.0f;
            gmax[j] = neg_inf;
            pos[j] = usable ? ((c == P - 1) ? 0u : c) : 0xFFFFFFFFu;
            const float d = gram[static_cast<size_t>(idx[j]) * stride + idx[j]]; // |row|^2, nominated
            bad |= !finite_f(d);
            diag_max = fmaxf(diag_max, finite_f(d) ? d : 0.0f);
            const float ta = __builtin_fabsf(t0[j]);
            t_abs = fmaxf(t_abs, finite_f(ta) ? ta : 0.0f);
        }
        diag_max = wave_max_f32_no_nan(diag_max);
        t_abs = wave_max_f32_no_nan(t_abs);
        const bool any_bad = __ballot(bad) != 0ull;
        // |nominated - reference| <= eps for every pair of this pool; |nominated MMR - reference MMR| <= delta (header)
        const float n2 = diag_max / (1.0f - eps_rel) * 1.0001f;
        const float eps = eps_rel * n2 * 1.0001f;
        const float delta = (lambda * eps + 2.3841858e-07f * (t_abs + lambda * (n2 + eps))) * 1.0001f;
        const float margin = 2.0f * delta;
        bool fallback = any_bad || !finite_f(margin);
        if (lane == 0) {
            s_order[0] = 0;
            s_eps = eps;
        }
        uint32_t n_rem = P - 1, last = 0;
        uint32_t n_amb_steps = 0, n_chains_steps = 0;
        while (!fallback && n_sel < k && n_rem > 0) {
            const float *g_last = gram + static_cast<size_t>(last) * stride;
            float sim[J], raw[J];
#pragma unroll
            for (int j = 0; j < J; ++j)
                sim[j] = g_last[idx[j]];
            float best_m = neg_inf;
            uint32_t best_pos = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                gmax[j] = fmaxf(gmax[j], sim[j]);       // (finite: the diagonal check vouches for every row)
                ms[j] = (sim[j] > ms[j]) ? sim[j] : ms[j];
                const float t1 = lambda * ms[j];
                const float m0 = t0[j] - t1;
                raw[j] = m0;
                const bool better = finite_f(m0) & ((m0 > best_m) | ((m0 == best_m) & (pos[j] < best_pos)));
                best_m = better ? m0 : best_m;
                best_pos = better ? pos[j] : best_pos;
            }
            const float wm = wave_max_f32_no_nan(best_m);
            if (wm == neg_inf) // no finite candidate left (:819-822)
                break;
            uint32_t wp = wave_min_u32(best_m == wm ? best_pos : 0xFFFFFFFFu);
            // how many candidates lie within the margin of the nominated winner?
            const float thr = wm - margin;
            uint32_t amb = 0;
#pragma unroll
            for (int j = 0; j < J; ++j)
                amb |= (finite_f(raw[j]) & (raw[j] >= thr)) ? (1u << j) : 0u;
            const unsigned long long any_lane = __ballot(amb != 0u);
            const bool certified = (__popcll(any_lane) == 1) && (__ballot((amb & (amb - 1u)) != 0u) == 0ull);
            if (!certified) {
                // count first: a crowd inside the margin (duplicated chunks) goes to the exact kernels
                uint32_t total = 0;
#pragma unroll
                for (int j = 0; j < J; ++j)
                    total += __popcll(__ballot((amb >> j) & 1u));
                if (total > kMaxAmbiguous) {
                    fallback = true;
                    break;
                }
                n_amb_steps++;
                float ex_best = neg_inf;
                uint32_t ex_pos = 0xFFFFFFFFu;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    unsigned long long mj = __ballot((amb >> j) & 1u);
                    while (mj) {
                        const int src = __builtin_ctzll(mj);
                        mj &= mj - 1ull;
                        const uint32_t c = static_cast<uint32_t>(src) + 64u * j;
                        const float g_star = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gmax[j]), src));
                        const float t0c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t0[j]), src));
                        const uint32_t pc = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(pos[j]), src));
                        const float ms_ex = exact_max_sim<F16>(gram + static_cast<size_t>(c) * stride, g_star, eps, s_order, n_sel, rows,
                                                               pitch_bytes, dim, list, c, lane);
                        n_chains_steps++;
                        const float t1 = lambda * ms_ex;
                        const float m_ex = t0c - t1;
                        const bool better = finite_f(m_ex) & ((m_ex > ex_best) | ((m_ex == ex_best) & (pc < ex_pos)));
                        ex_best = better ? m_ex : ex_best;
                        ex_pos = better ? pc : ex_pos;
                    }
                }
                if (ex_pos == 0xFFFFFFFFu) { // (cannot happen: the nominated winner is finite, so is its exact value)
                    fallback = true;
                    break;
                }
                wp = ex_pos;
            }
            uint32_t win = 0;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const bool hit = pos[j] == wp;
                win = hit ? lane + 64 * j + 1 : win;
                t0[j] = hit ? nan_f : t0[j];
                pos[j] = hit ? 0xFFFFFFFFu : (pos[j] == n_rem - 1 ? wp : pos[j]);
            }
            const unsigned long long ball = __ballot(win != 0);
            const int src = __builtin_ctzll(ball);
            last = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(win), src)) - 1;
            if (lane == 0) {
                out_order[n_sel] = last;
                s_order[n_sel] = last;
            }
            __builtin_amdgcn_wave_barrier(); // the other lanes read s_order in the exact evaluations of later steps
            n_sel++;
            n_rem--;
        }
        if (lane == 0) {
            s_nsel = fallback ? 0u : n_sel;
            *status = fallback ? 1u : 0u;
            if (!fallback) {
                out_order[0] = 0;
                out_mmr[0] = nan_f;
                *out_n = n_sel;
            }
            if (counters) {
                atomicAdd(&counters[0], n_amb_steps);
                atomicAdd(&counters[1], n_chains_steps);
                atomicAdd(&counters[2], fallback ? 1u : 0u);
            }
        }
    }
    __syncthreads();
    if (!want_values)
        return;
    // 3. the value the reference logs for pick t (:808-809 at the round that chose it): (1 - lambda) rel - lambda max_sim with
    // max_sim over the picks before it, from exact chains of the near-maximum picks only.  One wave per pick, round robin.
    const uint32_t n_done = s_nsel;
    const float eps = s_eps;
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t tpick = 1 + wave; tpick < n_done; tpick += 4) {
        const uint32_t c = s_order[tpick];
        const float *g_c = gram + static_cast<size_t>(c) * stride;
        float g_star = neg_inf;
        for (uint32_t base = 0; base < tpick; base += 64) {
            const uint32_t jj = base + lane;
            g_star = fmaxf(g_star, jj < tpick ? g_c[s_order[jj]] : neg_inf);
        }
        g_star = wave_max_f32_no_nan(g_star);
        const float ms_ex = exact_max_sim<F16>(g_c, g_star, eps, s_order, tpick, rows, pitch_bytes, dim, list, c, lane);
        const float t0c = one_minus * scores[c];
        const float t1 = lambda * ms_ex;
        if (lane == 0)
            out_mmr[tpick] = t0c - t1;
    }
}

} // namespace

// can the certified path serve pools of this index' rows?  binary16 rows whose width is a whole number of K chunks
bool mmr_nom_usable(uint32_t dim, int dtype)
{
    static const bool off = getenv("RLR_MMR_EXACT") != nullptr;
    return !off && dtype == RLR_F16 && dim % kKC == 0 && dim >= kKC;
}

float mmr_nom_eps_rel(uint32_t dim)
{
    // |mfma accumulation - true| <= dim * 2^-23 sum|p| (unit roundoff doubled: the matrix cores may truncate),
    // |reference chain - true| <= dim * 2^-24 sum|p|; 64 * 2^-24 on top
    return (3.0f * static_cast<float>(dim) + 64.0f) * 5.9604645e-8f;
}

hipError_t launch_gram_nom(const void *rows, uint32_t pitch16, uint32_t dim, const uint32_t *list, const uint32_t *sizes,
                           uint32_t P, float *gram, uint32_t n_pools, hipStream_t s)
{
    if (P == 0 || n_pools == 0)
        return hipSuccess;
    const uint32_t nt = (P + kT - 1) / kT;
    const uint32_t tiles = nt * (nt + 1) / 2;
    const uint32_t pools8 = (n_pools + 7) / 8 * 8;
    hipLaunchKernelGGL(gram_nom_kernel, dim3(pools8 * tiles), dim3(256), 0, s, static_cast<const unsigned char *>(rows), pitch16 * 16u,
                       dim, list, sizes, P, n_pools, tiles, gram);
    return hipGetLastError();
}

hipError_t launch_mmr_greedy_cert(const float *gram, const float *scores, uint32_t P, uint32_t k, float lambda, uint32_t *out_order,
                                  float *out_mmr, uint32_t *out_n, const uint32_t *sizes, uint32_t n_pools, const void *rows,
                                  uint32_t pitch16, uint32_t dim, const uint32_t *list, uint32_t *status, bool want_values,
                                  uint32_t *counters, hipStream_t s)
{
    if (P == 0 || P > 1024 || n_pools == 0)
        return hipErrorInvalidValue;
    const float eps_rel = mmr_nom_eps_rel(dim);
#define RLR_MMR_CERT(JV)                                                                                                   \
    hipLaunchKernelGGL((mmr_greedy_cert_kernel<JV, true>), dim3(n_pools), dim3(256), 0, s, gram, scores, P, k, lambda, eps_rel,     \
                       out_order, out_mmr, out_n, sizes, static_cast<const unsigned char *>(rows), pitch16 * 16u, dim, list,  \
                       status, want_values ? 1 : 0, counters)
    if (P <= 64)
        RLR_MMR_CERT(1);
    else if (P <= 128)
        RLR_MMR_CERT(2);
    else if (P <= 320)
        RLR_MMR_CERT(5);
    else if (P <= 512)
        RLR_MMR_CERT(8);
    else
        RLR_MMR_CERT(16);
#undef RLR_MMR_CERT
    return hipGetLastError();
}

} // namespace rlr
