// common.h -- shared host/device helpers for the gfx950 search path.
//
// Compile contract: the whole library is built with -ffp-contract=off, so `a*b+c`
// in source is a rounded product followed by a rounded add (the reference's
// arithmetic).  Fused multiply-adds appear only where the
// code spells fmaf()/__builtin_fmaf() (the wavefront-order candidate scan).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rlr {

constexpr int kWave = 64;            // gfx950 wavefront width
constexpr int kHistBits = 11;        // radix-select digit
constexpr int kHistBins = 1 << kHistBits;

// Order-preserving map f32 -> u32: a > b (as floats)  <=>  key(a) > key(b).
// NaN maps to 0 so it orders after every number ("NaN scores order last").
__host__ __device__ inline uint32_t score_key(float s)
{
    uint32_t b = __builtin_bit_cast(uint32_t, s);
    if ((b & 0x7FFFFFFFu) > 0x7F800000u)
        return 0u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__host__ __device__ inline float key_score(uint32_t k)
{
    if (k == 0u)
        return __builtin_bit_cast(float, 0x7FC00000u);
    uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __builtin_bit_cast(float, b);
}

// (score desc, row asc) as one descending u64.
__host__ __device__ inline uint64_t pack_result(float score, uint32_t row)
{
    return (static_cast<uint64_t>(score_key(score)) << 32) | static_cast<uint64_t>(0xFFFFFFFFu - row);
}

__host__ __device__ inline void unpack_result(uint64_t p, float *score, uint32_t *row)
{
    *score = key_score(static_cast<uint32_t>(p >> 32));
    *row = 0xFFFFFFFFu - static_cast<uint32_t>(p & 0xFFFFFFFFu);
}

#if defined(__HIPCC__)
// binary16 -> binary32, exact.
__device__ inline float h2f(uint16_t h)
{
    return static_cast<float>(__builtin_bit_cast(_Float16, h));
}

// DPP wavefront sum: result valid in lane 63 (and returned broadcast via readlane).
// row_shr 1/2/4/8 builds 16-lane suffix sums, row_bcast15/31 chain the four rows.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ inline float dpp_add(float v)
{
    int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, false);
    return v + __builtin_bit_cast(float, moved);
}

__device__ inline float wave_sum(float v)
{
    v = dpp_add<0x111, 0xF, 0xF>(v); // row_shr:1
    v = dpp_add<0x112, 0xF, 0xF>(v); // row_shr:2
    v = dpp_add<0x114, 0xF, 0xF>(v); // row_shr:4
    v = dpp_add<0x118, 0xF, 0xF>(v); // row_shr:8  -> lane 15 of each row = row sum
    v = dpp_add<0x142, 0xA, 0xF>(v); // row_bcast15 into rows 1,3
    v = dpp_add<0x143, 0xC, 0xF>(v); // row_bcast31 into rows 2,3 -> lane 63 = total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// DPP wavefront max (f32, no NaN inputs) / min (u32): same network as wave_sum, identity in `old`.
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_fmax(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(static_cast<int>(0xFF800000u), __builtin_bit_cast(int, v), CTRL,
                                                  ROW_MASK, 0xF, false);
    return __builtin_fmaxf(v, __builtin_bit_cast(float, moved));
}

__device__ inline float wave_max_f32(float v)
{
    v = dpp_fmax<0x111, 0xF>(v);
    v = dpp_fmax<0x112, 0xF>(v);
    v = dpp_fmax<0x114, 0xF>(v);
    v = dpp_fmax<0x118, 0xF>(v);
    v = dpp_fmax<0x142, 0xA>(v);
    v = dpp_fmax<0x143, 0xC>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// The same maximum for values that are never NaN: v_max_f32 with the DPP modifier on its own operand, one instruction per
// step (the builtin form above is a v_mov_b32_dpp plus a canonicalising v_max_f32 pair, because fmaxf must quiet NaNs).
// A lane whose DPP source is out of range is not written, i.e. keeps its value -- the identity of a maximum.
__device__ inline float wave_max_f32_no_nan(float v)
{
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int CTRL, int ROW_MASK>
__device__ inline uint32_t dpp_umin(uint32_t v)
{
    const int moved = __builtin_amdgcn_update_dpp(static_cast<int>(0xFFFFFFFFu), static_cast<int>(v), CTRL, ROW_MASK,
                                                  0xF, false);
    const uint32_t m = static_cast<uint32_t>(moved);
    return m < v ? m : v;
}

__device__ inline uint32_t wave_min_u32(uint32_t v)
{
    v = dpp_umin<0x111, 0xF>(v);
    v = dpp_umin<0x112, 0xF>(v);
    v = dpp_umin<0x114, 0xF>(v);
    v = dpp_umin<0x118, 0xF>(v);
    v = dpp_umin<0x142, 0xA>(v);
    v = dpp_umin<0x143, 0xC>(v);
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}
#endif

} // namespace rlr
