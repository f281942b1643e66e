// Operand splits for the 16-bit-input matrix cores: two f16 pieces for v_mfma_f32_32x32x16_f16 (rollout.hip,
// "network mode 1"; shared by the whole-rollout kernel and the fused minibatch update) and three bf16 pieces
// for v_mfma_f32_32x32x16_bf16 ("network mode 2": every f32 operand represented EXACTLY).
#pragma once
#include <stdint.h>

namespace ppo_net {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct Frag16 {
    uint32_t w[4];
};
using h8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __forceinline__ f32x16 mfma16(const Frag16 &a, const Frag16 &b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// (a0, a1) -> packed f16 pairs of the high pieces (round to nearest) and of the low pieces: the
// residual a - hi (exact in f32) rounded to f16, one v_fma_mixlo/mixhi_f16 each (src0 is read as
// the f16 half of `hi`, the other operands as f32).  hipcc does not form these from C (it packs
// the two residuals into v_pk_fma_f32 and converts separately: 5 instructions instead of 3).
__device__ __forceinline__ void split2(float a0, float a1, uint32_t &hi, uint32_t &lo)
{
    // round to nearest (v_cvt_pk_f16_f32, new in gfx950): the residual is at most half an f16 ulp, so
    // the low piece ends one bit further down than with v_cvt_pkrtz -- same instruction count
    using h2 = __attribute__((ext_vector_type(2))) _Float16;
    hi = __builtin_bit_cast(uint32_t, h2{(_Float16)a0, (_Float16)a1});
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(a1));
}

// ---- three bf16 pieces: x == b0 + b1 + b2 exactly ------------------------------------------------
// bf16 keeps f32's exponent range and 8 significand bits; rounding to nearest leaves a residual of at most half
// a bf16 ulp, i.e. 16 significant bits, the second piece leaves at most 8, which the third holds exactly
// (no underflow above 2^-110).  b0 = bf16(x) is one v_cvt_pk_bf16_f32 per PAIR of values; the residual
// r = x - b0 (exact in f32) is one v_dot2c_f32_bf16 per value: dot2((b0, b0'), (-1, 0)) + x reads the packed
// piece directly -- no unpacking (shift / mask) and no separate subtraction.  Seven instructions per pair
// (3.5 per value; the two-piece f16 split takes 1.5).
using bf8 = __attribute__((ext_vector_type(8))) __bf16;
using bf2 = __attribute__((ext_vector_type(2))) __bf16;

__device__ __forceinline__ f32x16 mfma16_bf16(const Frag16 &a, const Frag16 &b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

__device__ __forceinline__ uint32_t pack_bf16(float a0, float a1)
{
    return __builtin_bit_cast(uint32_t, bf2{(__bf16)a0, (__bf16)a1});      // v_cvt_pk_bf16_f32, round to nearest even
}

__device__ __forceinline__ void split3_bf16(float a0, float a1, uint32_t &p0, uint32_t &p1, uint32_t &p2)
{
    const bf2 take_lo = {(__bf16)-1.0f, (__bf16)0.0f}, take_hi = {(__bf16)0.0f, (__bf16)-1.0f};
    p0 = pack_bf16(a0, a1);
    const float r0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p0), take_lo, a0, false);
    const float r1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p0), take_hi, a1, false);
    p1 = pack_bf16(r0, r1);
    const float s0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p1), take_lo, r0, false);
    const float s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p1), take_hi, r1, false);
    p2 = pack_bf16(s0, s1);
}

}  // namespace ppo_net
