// tail.hip -- everything a single-query search does behind the scan, in two launches.
//
// The reference sorts every scored chunk and takes the first `initial_k` (/root/reference/src/rag_engine.rs:543-548).
// Here the scan (scan.hip) leaves the nominated scores (4 B per row) and the digit-1 radix histogram; what is left is
// integer work on those 4 bytes per row plus the reference-order re-score (`dot_product`, :1777-1779) of the
// handful of rows that survive.  Rounds 1-3 ran it as four dependent launches (digit-2 histogram, collect, re-score,
// sort: ~31 us of kernels and three launch boundaries, each of the first two a pass over the score array).  Now:
//
//   stage 1 (256-thread workgroups over the score array): every workgroup repeats the digit-1 bin search from the
//     8 KB histogram.  DIRECT mode -- at most `direct_max` scores sit in or above the bin of the k-th score, the
//     usual case up to a few million rows: the rows of its slice at or above (bin floor - guard band) are
//     appended to the candidate list (one slot reservation per workgroup) and re-scored on the spot by the workgroup
//     that found them -- coalesced row loads, products in parallel, one strict left-to-right chain per candidate
//     (staged_dot.h).  REFINE mode -- a crowded bin: the digit-2 histogram of that bin, as before.
//   stage 2 (1024-thread workgroups): DIRECT: workgroup 0 sorts the candidates in LDS and emits the best k; the others
//     only help clearing the histogram.  REFINE: digit-2 bin search, then collect + re-score exactly as above, and the
//     workgroup that finishes LAST (a counter in SelectState, agent-scope fences around it) sorts and emits.
//
// No workgroup ever waits for another one (no grid barrier): kernels of several searches may share the device.
// HBM/L2 traffic: the 4 n bytes of scores once (DIRECT) or twice (REFINE) + dim x elem bytes per candidate.
#include "common.h"
#include "kernels.h"
#include "lds_select.h"
#include "select_dev.h"
#include "pool_prepare.h"
#include "sort_emit.h"
#include "staged_dot.h"
#include "../../include/rlr_gpu.h"

#include <algorithm>

namespace rlr {

namespace {

constexpr uint32_t kLocalCap = 1024;    // candidates one workgroup keeps for its own re-score
constexpr uint32_t kSortBytes = 4096 * 8 + 2048 * 4; // sort_emit_body's LDS: 4096 keys + a 2048-bin histogram (>= kPoolLdsBytes)
static_assert(kPoolLdsBytes <= kSortBytes, "the pool preparation runs in the sort's LDS");

struct TailDev {
    const float *scores;
    uint32_t n;
    uint32_t *hist;
    SelectState *st;
    uint32_t k, cap;
    float two_eps;
    const float4 *rows;
    uint32_t pitch16, dim;
    const float *query;
    uint64_t *packed;
    uint64_t *out;
    uint64_t *meta;
    uint32_t unordered;
    uint32_t direct_max;
    uint32_t cpb;
    uint32_t has_pool;
    PoolArgs pool;
};

// The rows of this workgroup's slice whose nominated score key is >= key_lo: reserve their slots in the global candidate
// list, re-score them in reference order, store (exact score, row).  Called by all NT threads; s_mem = query + products.
// Returns the number of candidates found in the slice (uniform over the workgroup).
template <bool F16, int NT>
__device__ __forceinline__ uint32_t collect_rescore(const TailDev &a, uint32_t key_lo, float *s_mem)
{
    __shared__ uint32_t s_row[kLocalCap];
    __shared__ uint32_t s_cnt, s_base;
    const uint32_t tid = threadIdx.x;
    if (tid == 0)
        s_cnt = 0;
    __syncthreads();
    const uint32_t n4 = a.n / 4;
    const float4 *s4 = reinterpret_cast<const float4 *>(a.scores);
    // This workgroup's slice: one contiguous run of 16-byte score units (the grid may hold more workgroups than the pass
    // needs -- launch_tail_stage*: the re-score wants the candidates of a small corpus spread out).  Four independent
    // loads in flight per thread (a 10 M-row pass is ~10 loads per thread: one at a time behind the compare-and-append
    // it was latency-bound).
    const uint32_t chunk = (n4 + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = min(blockIdx.x * chunk, n4), hi = min(lo + chunk, n4);
    for (uint32_t i0 = lo + tid; i0 < hi; i0 += 4 * NT) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + u * NT;
            v[u] = s4[i < hi ? i : i0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + u * NT;
            if (i >= hi)
                break;
            const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (score_key(e[j]) >= key_lo) {
                    const uint32_t l = atomicAdd(&s_cnt, 1u);
                    if (l < kLocalCap)
                        s_row[l] = i * 4 + j;
                }
            }
        }
    }
    if (blockIdx.x == 0) // the up to three scores behind the last whole unit
        for (uint32_t i = n4 * 4 + tid; i < a.n; i += NT)
            if (score_key(a.scores[i]) >= key_lo) {
                const uint32_t l = atomicAdd(&s_cnt, 1u);
                if (l < kLocalCap)
                    s_row[l] = i;
            }
    __syncthreads();
    const uint32_t found = s_cnt;
    if (found == 0)
        return 0; // (uniform)
    if (tid == 0) {
        s_base = atomicAdd(&a.st->n_work, found);
        if (found > kLocalCap) // the surplus is lost here: the finish reports an overflow and the host takes the large-candidate path
            atomicOr(&a.st->flags, 1u);
    }
    const uint32_t q_floats = (a.dim + 7) & ~7u;
    float *s_q = s_mem;
    float *s_p = s_mem + q_floats;
    for (uint32_t i = tid; i < q_floats; i += NT)
        s_q[i] = i < a.dim ? a.query[i] : 0.0f;
    __syncthreads();
    const uint32_t base = s_base;
    const uint32_t cnt = min(found, kLocalCap);
    for (uint32_t g0 = 0; g0 < cnt; g0 += a.cpb) {
        const uint32_t c = min(a.cpb, cnt - g0);
        if (g0)
            __syncthreads(); // the chains of the previous group have left s_p
        const float sc = staged_reference_dot<F16, NT>(a.rows, a.pitch16, a.dim, s_q, s_p, s_row + g0, c, tid);
        const uint32_t slot = base + g0 + tid;
        if (tid < c && slot < a.cap) // agent-scope store: written through to where the other XCDs' workgroups read it (stage 2, REFINE)
            __hip_atomic_store(&a.packed[slot], pack_result(sc, s_row[g0 + tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return found;
}

template <bool F16>
__global__ __launch_bounds__(kSelThreads) void tail_stage1_kernel(TailDev a)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    uint32_t bin1, k2, in_bin;
    find_rank_bin(a.hist, a.k, &bin1, &k2, &in_bin);
    const uint32_t n_ge = a.k - k2 + in_bin; // scores in or above the digit-1 bin of the k-th
    const bool direct = n_ge <= a.direct_max;
    const uint32_t key_lo = band_floor_key(bin1, 0, a.two_eps);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        SelectState *st = a.st;
        st->k = a.k;
        st->bin1 = bin1;
        st->k2 = k2;
        st->cap = a.cap;
        st->mode = direct ? 1u : 2u;
        if (direct) {
            st->bin2 = 0;
            st->key_lo = key_lo;
        }
    }
    if (direct) {
        collect_rescore<F16, kSelThreads>(a, key_lo, s_mem);
        return;
    }
    // REFINE: digit 2 (key bits 20..10) of the scores whose digit 1 equals bin1
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_mem);
    for (int i = threadIdx.x; i < kHistBins; i += kSelThreads)
        s_hist[i] = 0;
    __syncthreads();
    const uint32_t stride = gridDim.x * kSelThreads;
    const uint32_t n4 = a.n / 4;
    const float4 *s4 = reinterpret_cast<const float4 *>(a.scores);
    for (uint32_t i = blockIdx.x * kSelThreads + threadIdx.x; i < n4; i += stride) {
        const float4 v = s4[i];
        const uint32_t k0 = score_key(v.x), k1 = score_key(v.y), k2_ = score_key(v.z), k3 = score_key(v.w);
        if ((k0 >> 21) == bin1) atomicAdd(&s_hist[(k0 >> 10) & (kHistBins - 1)], 1u);
        if ((k1 >> 21) == bin1) atomicAdd(&s_hist[(k1 >> 10) & (kHistBins - 1)], 1u);
        if ((k2_ >> 21) == bin1) atomicAdd(&s_hist[(k2_ >> 10) & (kHistBins - 1)], 1u);
        if ((k3 >> 21) == bin1) atomicAdd(&s_hist[(k3 >> 10) & (kHistBins - 1)], 1u);
    }
    for (uint32_t i = n4 * 4 + blockIdx.x * kSelThreads + threadIdx.x; i < a.n; i += stride) {
        const uint32_t k0 = score_key(a.scores[i]);
        if ((k0 >> 21) == bin1) atomicAdd(&s_hist[(k0 >> 10) & (kHistBins - 1)], 1u);
    }
    __syncthreads();
    uint32_t *g_hist2 = a.hist + kHistBins;
    for (int i = threadIdx.x; i < kHistBins; i += kSelThreads) {
        const uint32_t c = s_hist[i];
        if (c)
            atomicAdd(&g_hist2[i], c);
    }
}

// One workgroup, all candidates in place: publish the count, reset the tail's counters for the next query of this
// context, order and emit.  COHERENT: the candidates and counters were written by other workgroups of THIS launch (REFINE)
// and are read with agent-scope loads; in DIRECT mode the previous launch wrote them and plain loads do (three dependent
// write-through round trips less on the way to the sort: ~4 us of an 8 us stage).
// `pre` / `pre_n` / `pre_fl` (DIRECT only): packed[threadIdx.x], n_work and flags as the caller loaded them on entry, all
// at once with the mode word -- four dependent memory round trips in a row otherwise.
template <bool COHERENT>
__device__ __forceinline__ void tail_finish(const TailDev &a, float *s_mem, const uint64_t *pre = nullptr, uint32_t pre_n = 0,
                                            uint32_t pre_fl = 0)
{
    __shared__ uint32_t s_n;
    SelectState *st = a.st;
    if (threadIdx.x == 0) {
        const uint32_t n = COHERENT ? __hip_atomic_load(&st->n_work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : pre_n;
        const uint32_t fl = COHERENT ? __hip_atomic_load(&st->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : pre_fl;
        const uint32_t n_eff = (fl & 1u) ? max(n, a.cap + 1u) : n; // a local list overflowed: report a band overflow
        st->n_cand = n_eff;
        st->n_work = 0;
        st->done = 0;
        st->flags = 0;
        s_n = n_eff;
    }
    __syncthreads();
    const uint32_t n_eff = s_n;
    if (a.out) {
        uint64_t *s = reinterpret_cast<uint64_t *>(s_mem);
        uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_mem) + 4096 * 2;
        sort_emit_body<COHERENT>(a.packed, n_eff, a.cap, a.out, a.k, a.meta, a.unordered != 0, s, s_hist, pre);
    } else {
        if (threadIdx.x == 0 && a.meta)
            *a.meta = n_eff;
        if (a.has_pool) // a diversified search: the MMR pool straight from the candidates, in this launch (pool_prepare.h)
            pool_prepare_body<true, COHERENT>(a.packed, n_eff, a.cap, a.pool, reinterpret_cast<char *>(s_mem), pre);
    }
}

template <bool F16>
__global__ __launch_bounds__(1024) void tail_stage2_kernel(TailDev a)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    __shared__ uint32_t s_sel[3];
    __shared__ uint32_t s_last;
    const uint32_t tid = threadIdx.x;
    const uint32_t gid = blockIdx.x * 1024 + tid;
    SelectState *st = a.st;
    // workgroup 0 loads everything the DIRECT finish needs in one go (the candidate buffer always holds >= 1024 entries)
    uint64_t pre = 0;
    uint32_t pre_n = 0, pre_fl = 0;
    if (blockIdx.x == 0) {
        pre = a.packed[tid];
        pre_n = st->n_work;
        pre_fl = st->flags;
    }
    const uint32_t mode = st->mode;
    if (gid < kHistBins)
        a.hist[gid] = 0; // digit 1: stage 1 was its last reader
    if (mode == 1u) {
        if (blockIdx.x == 0)
            tail_finish<false>(a, s_mem, &pre, pre_n, pre_fl);
        return;
    }
    if (mode != 2u)
        return;
    lds_find_rank_bin_1024(a.hist + kHistBins, st->k2, s_sel);
    const uint32_t bin2 = s_sel[0];
    const uint32_t key_lo = band_floor_key(st->bin1, bin2, a.two_eps);
    collect_rescore<F16, 1024>(a, key_lo, s_mem);
    // The workgroup that arrives last orders and emits every workgroup's candidates.  The candidates were stored with
    // agent-scope (write-through) stores and are read back below with agent-scope loads, so no workgroup needs an
    // agent-scope FENCE -- which writes back and invalidates its XCD's whole L2 while the other workgroups still stream
    // scores through it: with one fence per wave the stage took 150 us at 10 M rows, with one per workgroup that had
    // candidates 36 us.  The barrier below waits for this workgroup's stores (vmcnt 0) before thread 0 counts it done.
    __syncthreads();
    if (tid == 0)
        s_last = atomicAdd(&st->done, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last)
        return;
    __threadfence(); // (one workgroup, once per query: acquire for everything it reads from here on)
    for (uint32_t i = tid; i < kHistBins; i += 1024)
        a.hist[kHistBins + i] = 0; // digit 2: every workgroup has read it
    if (tid == 0) {
        st->bin2 = bin2;
        st->key_lo = key_lo;
    }
    tail_finish<true>(a, s_mem);
}

bool tail_shape(uint32_t pitch16, uint32_t dim, int dtype, uint32_t *cpb_out, size_t *staging_out)
{
    const size_t q_bytes = static_cast<size_t>((dim + 7) & ~7u) * sizeof(float);
    const size_t row_bytes = q_bytes + 16; // f32 products of one candidate row (+16 B pad)
    if (pitch16 * (dtype == RLR_F16 ? 8u : 4u) < ((dim + 7) & ~7u))
        return false; // row pitch narrower than the 8-float rounding of dim (f32 rows, dim % 8 in 1..4)
    constexpr size_t budget = 56 * 1024; // + 4 KB of local candidate list + a few words: under the 64 KB a kernel gets by default
    uint32_t cpb = 8;
    while (cpb > 1 && q_bytes + cpb * row_bytes > budget)
        cpb >>= 1;
    if (q_bytes + cpb * row_bytes > budget)
        return false;
    *cpb_out = cpb;
    *staging_out = q_bytes + cpb * row_bytes;
    return true;
}

TailDev to_dev(const TailArgs &a, uint32_t cpb)
{
    TailDev d;
    d.scores = a.scores;
    d.n = a.n;
    d.hist = a.hist;
    d.st = a.st;
    d.k = a.k;
    d.cap = a.cap;
    d.two_eps = a.two_eps;
    d.rows = static_cast<const float4 *>(a.rows);
    d.pitch16 = a.pitch16;
    d.dim = a.dim;
    d.query = a.query;
    d.packed = a.packed;
    d.out = a.out;
    d.meta = a.meta;
    d.unordered = a.unordered ? 1u : 0u;
    d.direct_max = a.direct_max;
    d.cpb = cpb;
    d.has_pool = a.pool && !a.out ? 1u : 0u;
    d.pool = a.pool ? *a.pool : PoolArgs{};
    return d;
}

} // namespace

bool tail_fits(uint32_t pitch16, uint32_t dim, int dtype)
{
    uint32_t cpb;
    size_t staging;
    return tail_shape(pitch16, dim, dtype, &cpb, &staging);
}

hipError_t launch_tail_stage1(const TailArgs &a, hipStream_t s)
{
    uint32_t cpb;
    size_t staging;
    if (!tail_shape(a.pitch16, a.dim, a.dtype, &cpb, &staging))
        return hipErrorInvalidValue;
    const TailDev d = to_dev(a, cpb);
    const size_t lds = std::max<size_t>(staging, kHistBins * sizeof(uint32_t));
    // One 16-byte load per thread covers the pass; but the workgroups also re-score what they find, eight candidates at a
    // time, one 768-step chain each -- so a small corpus is spread over MORE workgroups than its scores need (partly idle
    // ones), until a few hundred candidates come out at a handful per workgroup: with 25 workgroups for 100 k rows the
    // ~1800 candidates of a text search's fetch took 85 us to re-score, ten groups in a row per workgroup.
    const uint32_t n4 = a.n / 4 + 1;
    uint32_t blocks = (n4 + kSelThreads - 1) / kSelThreads;
    blocks = std::max(blocks, std::min<uint32_t>(static_cast<uint32_t>(a.n_cu) * 2, (n4 + 31) / 32));
    blocks = std::max<uint32_t>(1, std::min<uint32_t>(blocks, static_cast<uint32_t>(a.n_cu) * 8));
    if (a.dtype == RLR_F16)
        hipLaunchKernelGGL(tail_stage1_kernel<true>, dim3(blocks), dim3(kSelThreads), lds, s, d);
    else
        hipLaunchKernelGGL(tail_stage1_kernel<false>, dim3(blocks), dim3(kSelThreads), lds, s, d);
    return hipGetLastError();
}

hipError_t launch_tail_stage2(const TailArgs &a, hipStream_t s)
{
    uint32_t cpb;
    size_t staging;
    if (!tail_shape(a.pitch16, a.dim, a.dtype, &cpb, &staging))
        return hipErrorInvalidValue;
    const TailDev d = to_dev(a, cpb);
    const size_t lds = std::max<size_t>(staging, kSortBytes);
    // at least two workgroups (2048 threads clear the digit-1 histogram); REFINE mode wants a pass over the scores.  One
    // 1024-thread workgroup per CU is what is resident at this kernel's register count: a second round of workgroups would
    // only start when the first has finished its bin search, pass and re-score.
    // (and, as in stage 1, at least one workgroup per 64 score units: the candidates of a small corpus spread out)
    // -- when many results are wanted: REFINE mode leaves at least k candidates to re-score, eight at a time per workgroup,
    // so a small grid is only safe for a small k (25 workgroups x 80 candidates each: 82 us at k = 1808).  A launch for a
    // few hundred results most likely finds DIRECT mode, where only workgroup 0 works, and stays small: 230 idle
    // 1024-thread workgroups cost ~2 us to dispatch and retire.
    const uint32_t n4 = a.n / 4 + 1;
    uint32_t blocks = (n4 + 1023) / 1024;
    if (a.k > 256 || a.k > a.direct_max)
        blocks = std::max(blocks, (n4 + 63) / 64);
    blocks = std::max<uint32_t>(2, std::min<uint32_t>(blocks, static_cast<uint32_t>(a.n_cu)));
    if (a.dtype == RLR_F16)
        hipLaunchKernelGGL(tail_stage2_kernel<true>, dim3(blocks), dim3(1024), lds, s, d);
    else
        hipLaunchKernelGGL(tail_stage2_kernel<false>, dim3(blocks), dim3(1024), lds, s, d);
    return hipGetLastError();
}

} // namespace rlr
