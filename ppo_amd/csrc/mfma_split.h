// Two-piece f16 operand splits for v_mfma_f32_32x32x16_f16 (the arithmetic is described in rollout.hip,
// "network mode 1"): shared by the whole-rollout kernel and the fused minibatch update.
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

}  // namespace ppo_net
