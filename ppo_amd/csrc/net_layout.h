// Sizes of the packed dense-network parameter buffer that k_rollout reads (rollout.hip documents the
// element order; pack.hip writes it).  Network: reference src/ppo.py:124-145 (64-64 tanh backbone(s),
// policy head, critic head).
#pragma once

namespace ppo_net {

constexpr int H = 64;  // hidden width of the reference's dense net (src/ppo.py:127-129)

// PPO_NET_F32: MFMA A-operand fragments, 1 bias k-step + K/2 input k-steps, [k-step][tile 2][lane 64]
__host__ __device__ constexpr int ksteps(int k_in) { return 1 + (k_in + 1) / 2; }
__host__ __device__ constexpr int backbone_floats(int obs) { return (ksteps(obs) + ksteps(H)) * 2 * 64; }
// heads (both modes): HL[4][2][16][64] + bias4[4]
constexpr int HEAD_FLOATS = 4 * 2 * 16 * 64 + 4;
// PPO_NET_F16_SPLIT, per backbone, in 32-bit words: W1[op 2][piece 2][lane 64][4], W2[j 4][op 2][piece 2][lane 64][4],
// B2[op 2][r 16][lane 64]
constexpr int BB16_W1 = 2 * 2 * 64 * 4, BB16_W2 = 4 * 2 * 2 * 64 * 4, BB16_B2 = 2 * 16 * 64;
constexpr int BB16_WORDS = BB16_W1 + BB16_W2 + BB16_B2;
// the same with P pieces per weight (P = 2: PPO_NET_F16_SPLIT, f16 pieces; P = 3: PPO_NET_BF16X3, bf16 pieces)
__host__ __device__ constexpr int bb16_w1(int P) { return 2 * P * 64 * 4; }
__host__ __device__ constexpr int bb16_w2(int P) { return 4 * 2 * P * 64 * 4; }
__host__ __device__ constexpr int bb16_words(int P) { return bb16_w1(P) + bb16_w2(P) + BB16_B2; }
static_assert(bb16_w1(2) == BB16_W1 && bb16_w2(2) == BB16_W2 && bb16_words(2) == BB16_WORDS, "two-piece sizes");

// hidden unit held by register r of output tile o in lane half h (the 32x32 MFMA D layout)
__host__ __device__ constexpr int unit(int o, int r, int h) { return 32 * o + (r & 3) + 8 * (r >> 2) + 4 * h; }

}  // namespace ppo_net
