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

// r = x - (piece `hi ? upper : lower` of the packed pair p), exact.  Both operands reach the builtin through
// registers the compiler cannot see through: hipcc 7.2 folds a CONSTANT bf16 pair -- the selector (-1, 0), or the
// packed pieces of a constant input such as the 1.0 that carries the bias -- into an inline operand (-1.0, 1.0),
// which the hardware does not read as that pair (tools/micro/bf16x3_probe.hip: every lower-half value wrong;
// in the kernel: wrong biases).  And v_dot2c_f32_bf16 written as inline asm loses the wait states the compiler
// inserts between the v_cvt_pk_bf16_f32 that makes p and the dot instruction that reads it (same probe).
struct Bf16Select {
    uint32_t lo, hi;       // bf16 pairs (-1, 0) and (0, -1), in scalar registers, made once per kernel
    __device__ __forceinline__ void init()
    {
        asm volatile("s_mov_b32 %0, 0xbf80" : "=s"(lo));
        asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(hi));
    }
};

__device__ __forceinline__ float residual_bf16(float x, uint32_t p, uint32_t sel)
{
    if (__builtin_constant_p(p)) asm("" : "+v"(p));        // (no instruction: hides a constant p from the folder)
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p), __builtin_bit_cast(bf2, sel), x, false);
}

__device__ __forceinline__ void split3_bf16(float a0, float a1, const Bf16Select &sel, uint32_t &p0, uint32_t &p1, uint32_t &p2)
{
    p0 = pack_bf16(a0, a1);
    const float r0 = residual_bf16(a0, p0, sel.lo), r1 = residual_bf16(a1, p0, sel.hi);
    p1 = pack_bf16(r0, r1);
    const float s0 = residual_bf16(r0, p1, sel.lo), s1 = residual_bf16(r1, p1, sel.hi);
    p2 = pack_bf16(s0, s1);
}

}  // namespace ppo_net
