// Packs the dense policy / value network into the layout the whole-rollout kernel reads -- gfx950.
//
// The reference keeps its network as torch.nn.Linear modules (src/ppo.py:124-145: Linear(obs,64)
// -tanh-Linear(64,64)-tanh backbone(s), policy head, critic head); k_rollout (rollout.hip) wants MFMA
// operand fragments.  One launch, one thread per output word, straight from the nn.Linear tensors
// (row-major [out, in] weights): it runs right behind optimizer.step() in the update, so a rollout
// never waits for a host-built gather plan (round 1: ~0.12 ms of small launches per update).
//
// A final block measures what decides whether PPO_NET_F16_SPLIT is as accurate as f32 for THIS
// parameter set (split_diag, three floats; w' = 2 log2(e) w is the value that is split):
//   [0] floor margin: the split keeps x = hi + lo in two f16 pieces, 22 significant bits, but the low
//       piece of a value below 2^-3 is an f16 subnormal, an ABSOLUTE error of 2^-25.  For a weight
//       tensor that floor costs sum_{|w'| < 2^-3} 2^-25 |x_k| against the mode's inherent
//       2^-22 sum |w' x_k|; [0] = min over the backbone weight tensors of 8 sum |w'| / #{|w'| < 2^-3},
//       the ratio of the two for inputs of equal size.  Measured (profiles/r02_net_precision.txt):
//       harmless down to 3e-4 -- with observations bounded by ~30 the floor stays below the
//       absolute error of the kernel's own tanh (2^-23), which bounds both network modes.
//   [1], [2] root-mean-square row norm of w' of the first / second backbone layer (max over the
//       backbones): the size of the pre-activations.  A 22-bit product is ~3x less accurate than an
//       f32 one; that shows in the outputs once the pre-activations are large enough for their
//       rounding error to exceed the tanh / output rounding floor, i.e. for LARGE weights.
// ppo_amd.ops.split_ok() turns the three numbers into the decision (thresholds calibrated and pinned
// by tests/test_gpu_net_precision.py).
#include "common.h"
#include "net_layout.h"

using namespace ppo_net;

namespace {

constexpr double TANH_SCALE = 2.0 * 1.4426950408889634;   // 2 log2(e): tanh(x) = 1 - 2 / (exp2(TANH_SCALE x) + 1)
constexpr int PACK_BLOCK = 256;

struct Backbone {
    const float *w1, *b1, *w2, *b2;   // nn.Linear layout: w1 [64, OBS], b1 [64], w2 [64, 64], b2 [64]
};

struct PackArgs {
    Backbone pi, va;
    const float *policy_w, *policy_b, *critic_w, *critic_b;   // [A, 64], [A], [1, 64], [1]
    float *out;
    float *split_diag;                // f32[3] or NULL
    int obs, A, joint, net_mode;
    int bb_words, total_words;
};

// x (already scaled, float64) -> its f16 hi / lo piece.  double -> float -> f16, both round-to-nearest:
// the same two roundings as torch's tensor.to(float16) on a float64 tensor.
__device__ __forceinline__ uint32_t f16_bits(double x) { return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)(float)x); }
__device__ __forceinline__ uint32_t split_piece(double v, int piece)
{
    const _Float16 hi = (_Float16)(float)v;
    if (piece == 0) return (uint32_t)__builtin_bit_cast(uint16_t, hi);
    return f16_bits(v - (double)(float)hi);
}

// x (already scaled, float64) -> bits of its bf16 piece 0, 1 or 2: b0 = bf16(x), b1 = bf16(x - b0),
// b2 = bf16(x - b0 - b1), each rounded to nearest even (double -> float -> bf16); the residuals are exact in
// float64.  b0 + b1 + b2 carries 24+ significant bits of x: the f32 value of the scaled weight, or better.
__device__ __forceinline__ double bf16_round(double x, uint32_t &bits)
{
    uint32_t u = __float_as_uint((float)x);
    u += 0x7fffu + ((u >> 16) & 1u);                   // round to nearest even on the upper 16 bits (finite inputs)
    bits = u >> 16;
    return (double)__uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ uint32_t split3_piece(double v, int piece)
{
    uint32_t bits;
    for (int q = 0; q < piece; ++q) v -= bf16_round(v, bits);
    bf16_round(v, bits);
    return bits;
}

// One 32-bit word of a backbone in the PPO_NET_F32 layout: F1[S1][2][64], F2[33][2][64]
__device__ float f32_backbone_word(const Backbone &b, int obs, int k)
{
    const int s1 = ksteps(obs) * 128;
    const bool second = k >= s1;
    if (second) k -= s1;
    const int s = k >> 7, o = (k >> 6) & 1, l = k & 63, half = l >> 5, row = 32 * o + (l & 31);
    const int K = second ? H : obs;
    const float *w = second ? b.w2 : b.w1, *bias = second ? b.b2 : b.b1;
    if (s == 0) return half == 0 ? bias[row] : 0.0f;
    int col;
    if (second) col = unit((s - 1) >> 4, (s - 1) & 15, half);
    else col = 2 * (s - 1) + half;
    return col < K ? w[row * K + col] : 0.0f;
}

// One 32-bit word (two 16-bit pieces) of a backbone in the PPO_NET_F16_SPLIT (P = 2, f16 pieces) or
// PPO_NET_BF16X3 (P = 3, bf16 pieces) layout
template <int P> __device__ uint32_t split_backbone_word(const Backbone &b, int obs, int k)
{
    constexpr int W1 = bb16_w1(P), W2 = bb16_w2(P);
    if (k >= W1 + W2) {                                // B2[op][r][lane] f32
        k -= W1 + W2;
        const int op = k >> 10, r = (k >> 6) & 15, l = k & 63;
        return __float_as_uint((float)((double)b.b2[unit(op, r, l >> 5)] * TANH_SCALE));
    }
    const auto piece_bits = [](double v, int piece) { return P == 2 ? split_piece(v, piece) : split3_piece(v, piece); };
    uint32_t word = 0;
    if (k < W1) {                                      // W1[op][piece][lane][4]
        const int wd = k & 3, l = (k >> 2) & 63, piece = (k >> 8) % P, op = (k >> 8) / P, row = 32 * op + (l & 31);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * wd + e;
            double v = 0.0;
            if (i < obs) v = (double)b.w1[row * obs + i] * TANH_SCALE;
            else if (i == obs) v = (double)b.b1[row] * TANH_SCALE;
            word |= piece_bits(v, piece) << (16 * e);
        }
        return word;
    }
    k -= W1;                                           // W2[j][op][piece][lane][4]
    const int wd = k & 3, l = (k >> 2) & 63, piece = (k >> 8) % P, op = ((k >> 8) / P) & 1, j = (k >> 8) / (2 * P);
    const int row = 32 * op + (l & 31);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int i = 2 * wd + e;
        const int col = unit(j >> 1, 8 * (j & 1) + i, l >> 5);
        word |= piece_bits((double)b.w2[row * H + col] * TANH_SCALE, piece) << (16 * e);
    }
    return word;
}

__device__ float head_word(const PackArgs &a, int k)
{
    if (k >= 4 * 2 * 16 * 64) {                        // bias4
        const int j = k - 4 * 2 * 16 * 64;
        if (j == 3) return a.critic_b[0];
        return j < a.A ? a.policy_b[j] : 0.0f;
    }
    const int l = k & 63, r = (k >> 6) & 15, o = (k >> 10) & 1, j = k >> 11;
    const int u = unit(o, r, l >> 5);
    if (j == 3) return a.critic_w[u];
    return j < a.A ? a.policy_w[j * H + u] : 0.0f;
}

// sum |w'|, #{|w'| < 2^-3} and sum w'^2 over n weights (block-wide); every thread returns the block totals
__device__ void tensor_sums(const float *w, int n, double &sum_abs, double &n_small, double &sum_sq)
{
    __shared__ double red[3][PACK_BLOCK / 64];
    double s = 0.0, c = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < n; i += PACK_BLOCK) {
        const double v = fabs((double)w[i] * TANH_SCALE);
        s += v;
        c += v < 0.125 ? 1.0 : 0.0;
        q += v * v;
    }
    s = wave_sum(s);
    c = wave_sum(c);
    q = wave_sum(q);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = c; red[2][threadIdx.x >> 6] = q; }
    __syncthreads();
    sum_abs = 0.0; n_small = 0.0; sum_sq = 0.0;
    for (int k = 0; k < PACK_BLOCK / 64; ++k) { sum_abs += red[0][k]; n_small += red[1][k]; sum_sq += red[2][k]; }
}

__global__ void __launch_bounds__(PACK_BLOCK) k_pack_params(PackArgs a)
{
    const int pack_blocks = (a.total_words + PACK_BLOCK - 1) / PACK_BLOCK;
    if ((int)blockIdx.x == pack_blocks) {              // the extra block: split-mode margin of this parameter set
        if (!a.split_diag) return;
        double margin = 1e30, g1 = 0.0, g2 = 0.0;
        const int nb = a.joint ? 1 : 2;
        for (int q = 0; q < nb; ++q) {
            const Backbone &b = q == 0 ? a.pi : a.va;
            double s, c, sq;
            tensor_sums(b.w1, H * a.obs, s, c, sq);
            if (c > 0.0) margin = fmin(margin, 8.0 * s / c);
            g1 = fmax(g1, sqrt(sq / H));                  // rms over the 64 rows of the row 2-norm
            tensor_sums(b.w2, H * H, s, c, sq);
            if (c > 0.0) margin = fmin(margin, 8.0 * s / c);
            g2 = fmax(g2, sqrt(sq / H));
        }
        if (threadIdx.x == 0) {
            a.split_diag[0] = (float)margin;
            a.split_diag[1] = (float)g1;
            a.split_diag[2] = (float)g2;
        }
        return;
    }
    const int k = blockIdx.x * PACK_BLOCK + threadIdx.x;
    if (k >= a.total_words) return;
    const int nbb = a.joint ? 1 : 2;
    uint32_t word;
    if (k < nbb * a.bb_words) {
        const Backbone &b = k < a.bb_words ? a.pi : a.va;
        const int kk = k < a.bb_words ? k : k - a.bb_words;
        word = a.net_mode == PPO_NET_F32 ? __float_as_uint(f32_backbone_word(b, a.obs, kk))
               : a.net_mode == PPO_NET_BF16X3 ? split_backbone_word<3>(b, a.obs, kk) : split_backbone_word<2>(b, a.obs, kk);
    } else {
        word = __float_as_uint(head_word(a, k - nbb * a.bb_words));
    }
    reinterpret_cast<uint32_t *>(a.out)[k] = word;
}

}  // namespace

extern "C" {

int ppo_rollout_pack_params(int kind, int joint, int net_mode, const float *pi_w1, const float *pi_b1,
                            const float *pi_w2, const float *pi_b2, const float *va_w1, const float *va_b1,
                            const float *va_w2, const float *va_b2, const float *policy_w, const float *policy_b,
                            const float *critic_w, const float *critic_b, float *params_out, float *split_diag,
                            void *stream)
{
    const int obs = ppo_env_obs_dim(kind);
    PPO_REQUIRE(obs > 0 && obs < 8 && obs % 2 == 0);
    PPO_REQUIRE(net_mode >= PPO_NET_F32 && net_mode <= PPO_NET_BF16X3);
    PPO_REQUIRE(pi_w1 && pi_b1 && pi_w2 && pi_b2 && policy_w && policy_b && critic_w && critic_b && params_out);
    PPO_REQUIRE(joint || (va_w1 && va_b1 && va_w2 && va_b2));
    PackArgs a{};
    a.pi = Backbone{pi_w1, pi_b1, pi_w2, pi_b2};
    a.va = Backbone{va_w1, va_b1, va_w2, va_b2};
    a.policy_w = policy_w; a.policy_b = policy_b; a.critic_w = critic_w; a.critic_b = critic_b;
    a.out = params_out; a.split_diag = split_diag;
    a.obs = obs; a.A = kind == PPO_ENV_CARTPOLE ? 2 : 3; a.joint = joint ? 1 : 0; a.net_mode = net_mode;
    a.bb_words = net_mode == PPO_NET_F32 ? backbone_floats(obs) : bb16_words(net_mode == PPO_NET_F16_SPLIT ? 2 : 3);
    a.total_words = a.bb_words * (joint ? 1 : 2) + HEAD_FLOATS;
    const unsigned grid = (unsigned)((a.total_words + PACK_BLOCK - 1) / PACK_BLOCK) + 1u;   // + the margin block
    hipLaunchKernelGGL(k_pack_params, dim3(grid), dim3(PACK_BLOCK), 0, (hipStream_t)stream, a);
    return ppo_launch_status();
}

}  // extern "C"
