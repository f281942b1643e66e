// Fused PPO minibatch update of the dense Agent with a Discrete head -- gfx950.
//
// What it replaces: one iteration of the reference's minibatch loop between the gathers and the
// optimiser (src/ppo.py:589-625): Agent.get_actions_and_values on the minibatch (:166-216, network
// :124-145: Linear(obs, 64)-tanh-Linear(64, 64)-tanh, policy head, critic head), the probability
// ratio (:595), compute_losses (:490-527), the loss assembly (:602-604 / :611-612, :619-620) and
// loss.backward().  PyTorch runs that as ~40 kernels over [B, 64] activation tensors (B = 2 M at
// BASELINE configs[1] with four minibatches: 3.7 ms, bound by activation traffic); here a sample's
// activations never leave the CU: forward, loss derivative, backward and the weight-gradient
// accumulation happen in one pass that reads ~40 B per sample.
//
// One wave works on 32 samples at a time (grid-stride).  All matrix products are exact-f32 MFMAs
// (v_mfma_f32_32x32x2_f32); a 32 x 32 result tile has lane = column, register r = row
// (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
//   forward    h1 = tanh(W1 x + b1), h2 = tanh(W2 h1 + b2): weights are the A operand (fragments in
//              LDS, built by the workgroup from the nn.Linear tensors when it starts), the 32 samples
//              are the columns, and a result register is the next layer's B operand as it is
//              (k order chosen to match, as in rollout.hip).  Heads: per-lane partial dot products
//              over the 32 units a lane holds, completed across the two lane halves.
//   loss       per sample (both lane halves redundantly): Categorical front end + clipped surrogate
//              / value / entropy terms and their derivatives (loss_core.h, shared with loss.hip).
//   backward   dz2 = (Whead^T dout) (1 - h2^2) on the VALU; dh1 = W2^T dz2 again register-chained
//              (transposed fragments in LDS); dz1 = dh1 (1 - h1^2).
//   gradients  dW2 = sum_s dz2 h1^T etc. need the SAMPLES as the MFMA k dimension, i.e. operands with
//              lane = unit: the wave transposes dz2 / h1 / dz1 / h2 through its own LDS staging tiles
//              ([unit][sample], padded) and accumulates the 32 x 32 gradient tiles over all of its
//              samples in MFMA accumulators; bias gradients are column sums kept per lane.
// The four waves of a workgroup then add their partial gradients in LDS, in the parameters' own layout,
// one row per workgroup goes to the workspace, and k_update_reduce adds the rows in a fixed order in
// f64 (deterministic) and finishes the loss scalars.  The minibatch is gathered on the fly: a sample's
// inputs are fetched through idx one tile ahead of their use.
#include <type_traits>
#include "common.h"
#include "loss_core.h"
#include "net_layout.h"
#include "mfma_split.h"

using namespace ppo_net;
using namespace ppo_loss;

namespace {

#define UPD_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32((a_), (b_), (c_), 0, 0, 0)
// 16 x 16 tiles (k = 4 per instruction, half the cycles of the 32 x 32 form) for the two skinny gradient
// products, dW1 (<= 7 useful columns) and the head rows (<= 4 useful rows): lane l supplies
// A[row = l & 15][k = l >> 4] and B[k = l >> 4][col = l & 15]; result register r of lane l is
// D[row = 4 (l >> 4) + r][col = l & 15].
using f32x4 = __attribute__((ext_vector_type(4))) float;
#define UPD_MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x4f32((a_), (b_), (c_), 0, 0, 0)

constexpr int UPD_BLOCK = 256;            // 4 waves, one per SIMD
constexpr int UPD_MAX_BLOCKS = 256;       // one workgroup per CU (LDS: ~110 KB)
constexpr int SPAD = 33;                  // staging row: 32 samples + 1 (conflict-free transposed reads)
constexpr int NSCAL = 5;                  // sum m, sum vterm, sum clipped, sum kl, sum H

struct Net {                              // one backbone + the heads that hang on it (nn.Linear layouts)
    const float *w1, *b1, *w2, *b2;       // [64, OBS], [64], [64, 64], [64]
    const float *policy_w, *policy_b;     // [A, 64], [A]   (NULL unless DO_P)
    const float *critic_w, *critic_b;     // [1, 64], [1]   (NULL unless DO_V)
};

struct UpdArgs {
    Net net;
    const long *idx;                      // [B] rows of the arrays below that form the minibatch, or NULL (rows 0 .. B-1)
    const float *states;                  // [rows, OBS]
    const int *actions;                   // [rows]
    const float *old_lp, *old_v, *adv;    // [rows]
    const double *stats3;                 // advantage statistics of the minibatch (ppo_adv_stats), or NULL
    float lo, hi, epsf;
    float c_pi, c_v, c_ent;               // gradient of c_pi policy_loss + c_v value_loss - c_ent entropy
    long B;
    int norm_adv, clip_va;
    float *partials;                      // [blocks][P] per-workgroup gradient partials, parameter order (GradLayout)
    double *scal;                         // [waves][NSCAL]
    int P;
};

// Offsets of the gradient tensors inside one partial row / the output buffer (floats):
//   w1 [64, OBS], b1 [64], w2 [64, 64], b2 [64], policy_w [A, 64], policy_b [A], critic_w [64], critic_b [1]
template <int OBS, int A, bool DO_P, bool DO_V> struct GradLayout {
    static constexpr int W1 = 0, B1 = W1 + H * OBS, W2 = B1 + H, B2 = W2 + H * H, PW = B2 + H;
    static constexpr int PB = PW + (DO_P ? A * H : 0), CW = PB + (DO_P ? A : 0), CB = CW + (DO_V ? H : 0);
    static constexpr int P = CB + (DO_V ? 1 : 0);
};

__device__ __forceinline__ f32x16 zero16()
{
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}

// tanh as in rollout.hip: 1 - 2 / (exp(2x) + 1) on the exp2 / rcp units (absolute error < 2e-7)
__device__ __forceinline__ float fast_tanh(float x)
{
    const float t = __builtin_amdgcn_exp2f(2.8853900817779268f * x);
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
}

// two at a time: the multiply, the add and the fma as packed-f32 instructions (one issue slot for two values);
// same operations, same results
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 fast_tanh2(f32x2 x)
{
    const f32x2 y = x * 2.8853900817779268f;
    f32x2 e = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
    e += 1.0f;
    const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    return __builtin_elementwise_fma(f32x2{-2.0f, -2.0f}, r, f32x2{1.0f, 1.0f});
}
__device__ __forceinline__ f32x2 one_minus_sq(f32x2 h) { return __builtin_elementwise_fma(-h, h, f32x2{1.0f, 1.0f}); }   // tanh'

// sum of x over the lane and its partner in the other half (lane ^ 32), in both
__device__ __forceinline__ float sum_halves(float x)
{
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));   // a: lower-half value, b: upper-half value
    return a + b;
}

// max over the wave of the bit patterns of non-negative floats (they order like the floats), as a wave-uniform
// value: DPP steps inside each row of 16 lanes (no LDS round trips, unlike __shfl_xor), then the four rows through
// v_readlane
__device__ __forceinline__ uint32_t wave_max_bits(uint32_t v)
{
    const auto step = [](uint32_t x, auto ctrl) {
        const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, decltype(ctrl)::value, 0xF, 0xF, true);
        return x > y ? x : y;
    };
    v = step(v, std::integral_constant<int, 0xB1>{});     // quad_perm [1, 0, 3, 2]
    v = step(v, std::integral_constant<int, 0x4E>{});     // quad_perm [2, 3, 0, 1]
    v = step(v, std::integral_constant<int, 0x141>{});    // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{});    // row_mirror
    const uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// f16-split form of a register-chained 64 x 64 layer (F16 = true): the 32 registers a lane holds of a
// [64 units][32 samples] tile -> the four k-steps' B operands, high and low pieces
__device__ __forceinline__ void split_tile(const f32x16 (&v)[2], Frag16 (&bh)[4], Frag16 (&bl)[4])
{
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int r = 0; r < 16; r += 2) split2(v[o][r], v[o][r + 1], bh[2 * o + (r >> 3)].w[(r & 7) >> 1], bl[2 * o + (r >> 3)].w[(r & 7) >> 1]);
}
// acc[row tile] += W (fragments at `frag`: [k-step 4][row tile 2][piece 2][lane 64] x 16 B) times the split
// operand: xh wl + xl wh first, xh wh last (small terms first), the two row tiles' chains interleaved
__device__ __forceinline__ void chain16(const float *frag, int lane, const Frag16 (&bh)[4], const Frag16 (&bl)[4], f32x16 (&acc)[2])
{
    const uint4 *f = reinterpret_cast<const uint4 *>(frag);
    const auto get = [&](int j, int o, int pc) {
        const uint4 x = f[((j * 2 + o) * 2 + pc) * 64 + lane];
        return Frag16{{x.x, x.y, x.z, x.w}};
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 0; o < 2; ++o) acc[o] = mfma16(get(j, o, 1), bh[j], acc[o]);
#pragma unroll
        for (int o = 0; o < 2; ++o) acc[o] = mfma16(get(j, o, 0), bl[j], acc[o]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int o = 0; o < 2; ++o) acc[o] = mfma16(get(j, o, 0), bh[j], acc[o]);
}

template <int OBS, int A, bool DO_P, bool DO_V, bool F16>
__global__ void __launch_bounds__(UPD_BLOCK) k_update(UpdArgs a)
{
    using L = GradLayout<OBS, A, DO_P, DO_V>;
    constexpr int S1 = ksteps(OBS);                    // 1 bias k-step + OBS / 2
    constexpr int NHEAD = 4;                           // head rows: 0..A-1 logits, 3 = critic
    static_assert(OBS % 2 == 0 && OBS < 8 && A <= 3, "layouts assume an even input width < 8 and <= 3 actions");
    // ---- LDS (one array, carved by hand: the gradient rows of the four waves are added in it at the
    // end, over everything else): weight fragments shared by the block, then the per-wave staging tiles
    constexpr int NW = UPD_BLOCK / 64;
    constexpr int OFF_F1 = 0, OFF_F2 = OFF_F1 + S1 * 128, OFF_F2T = OFF_F2 + 33 * 128, OFF_HW = OFF_F2T + 32 * 128;
    // per-wave staging: two f32 tiles [unit 64][sample 32 + 1]; the f16 form lays four tiles of packed piece pairs
    // ([unit pair 32][sample 32 + 4]: dz2 high / low, h1 high / low) over the same words
    constexpr int PROW = 36, PTILE = 32 * PROW;
    constexpr int STG_WAVE = F16 ? 4 * PTILE : 2 * H * SPAD;
    static_assert(STG_WAVE >= 2 * H * SPAD && STG_WAVE % 4 == 0, "the f32 staging tiles live in the same words");
    constexpr int OFF_HB = OFF_HW + NHEAD * H, OFF_SA = OFF_HB + 4;
    constexpr int OFF_SX = OFF_SA + NW * STG_WAVE, LDS_FLOATS = OFF_SX + NW * 8 * SPAD;
    static_assert(LDS_FLOATS >= NW * L::P, "the waves' gradient rows must fit for the final addition");
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float(*F1)[2][64] = reinterpret_cast<float(*)[2][64]>(lds + OFF_F1);     // layer 1, A operand: [k-step][row tile][lane]
    float(*F2)[2][64] = reinterpret_cast<float(*)[2][64]>(lds + OFF_F2);     // layer 2
    float(*F2T)[2][64] = reinterpret_cast<float(*)[2][64]>(lds + OFF_F2T);   // layer 2 transposed (dh1 = W2^T dz2)
    float(*HW)[H] = reinterpret_cast<float(*)[H]>(lds + OFF_HW);             // head rows, natural order
    float *hbias = lds + OFF_HB;
    float(*SX)[8][SPAD] = reinterpret_cast<float(*)[8][SPAD]>(lds + OFF_SX);   // inputs (+ a row of ones at OBS) / head derivatives
    const Net &n = a.net;
    for (int k = threadIdx.x; k < S1 * 128; k += UPD_BLOCK) {
        const int s = k >> 7, o = (k >> 6) & 1, l = k & 63, row = 32 * o + (l & 31);
        float v;
        if (s == 0) v = l < 32 ? n.b1[row] : 0.0f;
        else v = n.w1[row * OBS + 2 * (s - 1) + (l >> 5)];
        F1[s][o][l] = v;
    }
    if constexpr (!F16) {
        for (int k = threadIdx.x; k < 33 * 128; k += UPD_BLOCK) {
            const int s = k >> 7, o = (k >> 6) & 1, l = k & 63, row = 32 * o + (l & 31);
            float v;
            if (s == 0) v = l < 32 ? n.b2[row] : 0.0f;
            else v = n.w2[row * H + unit((s - 1) >> 4, (s - 1) & 15, l >> 5)];
            F2[s][o][l] = v;
        }
        for (int k = threadIdx.x; k < 32 * 128; k += UPD_BLOCK) {
            const int s = k >> 7, o = (k >> 6) & 1, l = k & 63;
            // A[row = h1 unit 32 o + (l & 31)][k = h2 unit unit(s >> 4, s & 15, l >> 5)] = W2[h2 unit][h1 unit]
            F2T[s][o][l] = n.w2[unit(s >> 4, s & 15, l >> 5) * H + 32 * o + (l & 31)];
        }
    } else {
        // f16-split A operands of v_mfma_f32_32x32x16_f16 (layout of rollout.hip's W2): [k-step j 4][row tile 2]
        // [piece 2][lane 64] x 8 f16; element i of lane l in k-step j is hidden unit unit(j >> 1, 8 (j & 1) + i, l >> 5)
        // of the k dimension -- the order in which a 32 x 32 result tile's registers hold them.  F2H: W2 itself
        // (rows = h2 units), F2TH: its transpose (rows = h1 units, dh1 = W2^T dz2).
        for (int k = threadIdx.x; k < 2 * 4 * 2 * 64 * 4; k += UPD_BLOCK) {
            const int q = k & 3, l = (k >> 2) & 63, o = (k >> 8) & 1, j = (k >> 9) & 3, tr = k >> 11;
            const int row = 32 * o + (l & 31);
            uint32_t hi = 0, lo = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = 2 * q + e, ku = unit(j >> 1, 8 * (j & 1) + i, l >> 5);
                const float w = tr ? n.w2[ku * H + row] : n.w2[row * H + ku];
                const _Float16 h = (_Float16)w, r = (_Float16)(w - (float)h);
                hi |= (uint32_t)__builtin_bit_cast(unsigned short, h) << (16 * e);
                lo |= (uint32_t)__builtin_bit_cast(unsigned short, r) << (16 * e);
            }
            uint32_t *dst = reinterpret_cast<uint32_t *>(lds + (tr ? OFF_F2T : OFF_F2));
            dst[(((j * 2 + o) * 2 + 0) * 64 + l) * 4 + q] = hi;
            dst[(((j * 2 + o) * 2 + 1) * 64 + l) * 4 + q] = lo;
        }
        // b2 as the C operand of the first k-step: [lane half][row tile][register]
        for (int k = threadIdx.x; k < 64; k += UPD_BLOCK) lds[OFF_F2 + 4096 + k] = n.b2[unit((k >> 4) & 1, k & 15, k >> 5)];
    }
    for (int k = threadIdx.x; k < NHEAD * H; k += UPD_BLOCK) {
        const int j = k >> 6, u = k & 63;
        float v = 0.0f;
        if (DO_P && j < A) v = n.policy_w[j * H + u];
        if (DO_V && j == 3) v = n.critic_w[u];
        HW[j][u] = v;
    }
    if (threadIdx.x < NHEAD) {
        float v = 0.0f;
        if (DO_P && (int)threadIdx.x < A) v = n.policy_b[threadIdx.x];
        if (DO_V && threadIdx.x == 3) v = n.critic_b[0];
        hbias[threadIdx.x] = v;
    }
    __syncthreads();                                   // the only barrier: waves are independent from here on

    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31, wv = threadIdx.x >> 6;
    const float one = half == 0 ? 1.0f : 0.0f;
    float *stage = lds + OFF_SA + wv * STG_WAVE;
    float(*sa)[SPAD] = reinterpret_cast<float(*)[SPAD]>(stage);               // staging: [unit][sample]
    float(*sb)[SPAD] = reinterpret_cast<float(*)[SPAD]>(stage + H * SPAD);
    uint32_t *pst = reinterpret_cast<uint32_t *>(stage);                      // f16 form: tiles 0 / 1 dz2 high / low, 2 / 3 h1 high / low
    const uint32_t psel = (lane & 1) ? 0x07060302u : 0x05040100u;              // v_perm: the odd / even unit's half of two staged words
    // k-steps' pieces (a lane's registers of a [64 units][32 samples] tile, two adjacent units per word) -> staging tile
    const auto put_pieces = [&](int tile, const Frag16 (&b)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = j >> 1, r = 8 * (j & 1) + 2 * q;
                pst[tile * PTILE + (unit(o, r, 0) / 2 + 2 * half) * PROW + col] = b[j].w[q];
            }
    };
    // MFMA operand with the samples as k: row (or column) = unit 32 t + (lane & 31), elements = samples 16 m + 8 half + 0..7
    const auto get_pieces = [&](int tile, int t, int m) {
        const uint4 *src = reinterpret_cast<const uint4 *>(pst + tile * PTILE + (16 * t + (col >> 1)) * PROW + 16 * m + 8 * half);
        const uint4 x = src[0], y = src[1];
        Frag16 f;
        f.w[0] = __builtin_amdgcn_perm(x.y, x.x, psel);
        f.w[1] = __builtin_amdgcn_perm(x.w, x.z, psel);
        f.w[2] = __builtin_amdgcn_perm(y.y, y.x, psel);
        f.w[3] = __builtin_amdgcn_perm(y.w, y.z, psel);
        return f;
    };
    int Ecur = 0, Emax = 0;                        // f16 form: biased exponent the dW2 accumulators are scaled for (0: nothing yet)
    float(*sx)[SPAD] = SX[wv];
    const LossConsts c = loss_consts(a.stats3, a.norm_adv, a.clip_va, a.lo, a.hi, a.epsf, a.B);

    // gradient accumulators over all of this wave's samples
    f32x16 gW2[2][2] = {{zero16(), zero16()}, {zero16(), zero16()}};   // [row tile: h2 unit][col tile: h1 unit]
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 gW1[4] = {zero4, zero4, zero4, zero4};                       // [16-row tile: h1 unit], cols: input i (OBS = bias)
    f32x4 gWh[4] = {zero4, zero4, zero4, zero4};                       // rows: head j, [16-col tile: h2 unit]
    f32x2 gb2[2][8];                                                   // column sums of dz2, this lane's samples (register pairs)
    float gbh[NHEAD] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int r = 0; r < 8; ++r) gb2[o][r] = f32x2{0.0f, 0.0f};
    double acc[NSCAL] = {0.0, 0.0, 0.0, 0.0, 0.0};

    const long tiles = (a.B + 31) / 32;
    const long wave_id = (long)blockIdx.x * (UPD_BLOCK / 64) + wv, nwaves = (long)gridDim.x * (UPD_BLOCK / 64);
    // A sample's inputs are fetched one tile ahead (and its row index two tiles ahead when the minibatch is
    // given by idx): nothing in a tile's work touches global memory, so the loads fly under its MFMAs and
    // the separate gather launch of the unfused path (and its write + re-read of the minibatch) is gone.
    struct In {
        float x[OBS];
        int act;
        float old_lp, old_v, adv;
    };
    const auto row_of = [&](long t) -> long {          // source row of this lane's sample of tile t (clamped)
        long i = t * 32 + col;
        if (i >= a.B) i = a.B - 1;
        return a.idx ? a.idx[i] : i;
    };
    const auto fetch = [&](long row) {
        In v;
        if (OBS == 4) {
            const float4 q = *reinterpret_cast<const float4 *>(a.states + row * 4);
            v.x[0] = q.x; v.x[1] = q.y; v.x[2] = q.z; v.x[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < OBS; k += 2) {
                const float2 q = *reinterpret_cast<const float2 *>(a.states + row * OBS + k);
                v.x[k] = q.x; v.x[k + 1] = q.y;
            }
        }
        v.act = a.actions[row];
        v.old_lp = a.old_lp[row]; v.old_v = a.old_v[row]; v.adv = a.adv[row];
        return v;
    };
    In nxt = fetch(row_of(wave_id < tiles ? wave_id : 0));
    long row2 = row_of(wave_id + nwaves < tiles ? wave_id + nwaves : 0);
    for (long t = wave_id; t < tiles; t += nwaves) {
        const bool valid = t * 32 + col < a.B;
        const In cur = nxt;
        nxt = fetch(row2);
        row2 = row_of(t + 2 * nwaves < tiles ? t + 2 * nwaves : 0);
        float x[OBS];
#pragma unroll
        for (int k = 0; k < OBS; ++k) x[k] = cur.x[k];
        const int act = cur.act;
        const float old_lp = cur.old_lp, old_v = cur.old_v, adv = cur.adv;

        // ---- forward
        f32x16 h1[2], h2[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) h1[o] = UPD_MFMA(F1[0][o][lane], one, zero16());
#pragma unroll
        for (int s = 1; s < S1; ++s) {
            const float b = half ? x[2 * (s - 1) + 1] : x[2 * (s - 1)];
#pragma unroll
            for (int o = 0; o < 2; ++o) h1[o] = UPD_MFMA(F1[s][o][lane], b, h1[o]);
        }
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 t = fast_tanh2(f32x2{h1[o][r], h1[o][r + 1]});
                h1[o][r] = t[0]; h1[o][r + 1] = t[1];
            }
        if constexpr (!F16) {
#pragma unroll
            for (int o = 0; o < 2; ++o) h2[o] = UPD_MFMA(F2[0][o][lane], one, zero16());
#pragma unroll
            for (int k = 0; k < 32; ++k) {
#pragma unroll
                for (int o = 0; o < 2; ++o) h2[o] = UPD_MFMA(F2[1 + k][o][lane], h1[k >> 4][k & 15], h2[o]);
            }
        } else {
            Frag16 bh[4], bl[4];
            split_tile(h1, bh, bl);
            put_pieces(2, bh);                          // for dW2 below: the previous tile is done with these words
            put_pieces(3, bl);
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 b = *reinterpret_cast<const float4 *>(lds + OFF_F2 + 4096 + (half * 2 + o) * 16 + 4 * r4);
                    h2[o][4 * r4] = b.x; h2[o][4 * r4 + 1] = b.y; h2[o][4 * r4 + 2] = b.z; h2[o][4 * r4 + 3] = b.w;
                }
            chain16(lds + OFF_F2, lane, bh, bl, h2);
        }
        float p[NHEAD] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int u0 = unit(o, 4 * r4, 0);             // units 4 r4 .. 4 r4 + 3 of this lane half are adjacent
                float hv[4];
                {
                    const f32x2 h01 = fast_tanh2(f32x2{h2[o][4 * r4], h2[o][4 * r4 + 1]});
                    const f32x2 h23 = fast_tanh2(f32x2{h2[o][4 * r4 + 2], h2[o][4 * r4 + 3]});
                    hv[0] = h2[o][4 * r4] = h01[0]; hv[1] = h2[o][4 * r4 + 1] = h01[1];
                    hv[2] = h2[o][4 * r4 + 2] = h23[0]; hv[3] = h2[o][4 * r4 + 3] = h23[1];
                }
#pragma unroll
                for (int j = 0; j < NHEAD; ++j) {
                    if ((DO_P && j < A) || (DO_V && j == 3)) {
                        const float4 w = *reinterpret_cast<const float4 *>(&HW[j][u0 + 4 * half]);
                        p[j] = __builtin_fmaf(hv[0], w.x, p[j]);
                        p[j] = __builtin_fmaf(hv[1], w.y, p[j]);
                        p[j] = __builtin_fmaf(hv[2], w.z, p[j]);
                        p[j] = __builtin_fmaf(hv[3], w.w, p[j]);
                    }
                }
            }
        float out[NHEAD] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NHEAD; ++j)
            if ((DO_P && j < A) || (DO_V && j == 3)) out[j] = sum_halves(p[j]) + hbias[j];

        // ---- loss and its derivative wrt the head outputs (both lane halves compute their sample's)
        float dout[NHEAD] = {0.f, 0.f, 0.f, 0.f};
        float rho = 1.0f, H_ent = 0.0f;
        if (DO_P) {
            float lg[A];
#pragma unroll
            for (int k = 0; k < A; ++k) lg[k] = out[k];
            Categorical<A> cat;
            cat.from_logits(lg, act);
            rho = expf(cat.new_lp - old_lp);                                   // :595
            H_ent = cat.H;
            const SampleOut o = loss_sample(c, rho, adv, old_v, DO_V ? out[3] : old_v);
            const float g_lp = o.g_rho * rho;                                  // d policy_loss / d new_lp
#pragma unroll
            for (int k = 0; k < A; ++k) {
                const float g_pl = g_lp * ((k == act ? 1.0f : 0.0f) - cat.pr[k]);
                const float g_en = -cat.pr[k] * (cat.lp[k] + cat.H) * c.invB;   // d mean(H) / d logit_k
                dout[k] = a.c_pi * g_pl - a.c_ent * g_en;
            }
            if (DO_V) dout[3] = a.c_v * o.g_v;
            if (valid && half == 0) {
                acc[0] += (double)o.m;
                acc[2] += (double)o.clipped;
                acc[3] += (double)o.kl;
                acc[4] += (double)cat.H;
                if (DO_V) acc[1] += (double)o.vterm;
            }
        } else {
            const SampleOut o = loss_sample(c, 1.0f, adv, old_v, out[3]);
            dout[3] = a.c_v * o.g_v;
            if (valid && half == 0) acc[1] += (double)o.vterm;
        }
        (void)H_ent;
        if (!valid) {
#pragma unroll
            for (int j = 0; j < NHEAD; ++j) dout[j] = 0.0f;
        }
        if (half == 0) {
#pragma unroll
            for (int j = 0; j < NHEAD; ++j) gbh[j] += dout[j];
        }

        // ---- dz2 = (Whead^T dout) (1 - h2^2)
        f32x16 dz2[2];
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int u0 = unit(o, 4 * r4, 0) + 4 * half;
                f32x2 d[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
#pragma unroll
                for (int j = 0; j < NHEAD; ++j) {
                    if ((DO_P && j < A) || (DO_V && j == 3)) {
                        const float4 w = *reinterpret_cast<const float4 *>(&HW[j][u0]);
                        const f32x2 dj = {dout[j], dout[j]};
                        d[0] = __builtin_elementwise_fma(f32x2{w.x, w.y}, dj, d[0]);
                        d[1] = __builtin_elementwise_fma(f32x2{w.z, w.w}, dj, d[1]);
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = 4 * r4 + 2 * q;
                    const f32x2 z = d[q] * one_minus_sq(f32x2{h2[o][r], h2[o][r + 1]});
                    dz2[o][r] = z[0]; dz2[o][r + 1] = z[1];
                    gb2[o][r >> 1] += z;
                }
            }

        f32x16 dz1[2] = {zero16(), zero16()};
        if constexpr (!F16) {
            // ---- dW2 += dz2 h1^T: transpose both through the staging tiles, samples become the k dimension
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sa[unit(o, r, half)][col] = dz2[o][r];
                    sb[unit(o, r, half)][col] = h1[o][r];
                }
            __builtin_amdgcn_wave_barrier();           // (compiler ordering only: a wave's LDS operations execute in order)
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const int s = 2 * j + half;
                const float a0 = sa[col][s], a1 = sa[32 + col][s], b0 = sb[col][s], b1 = sb[32 + col][s];
                gW2[0][0] = UPD_MFMA(a0, b0, gW2[0][0]);
                gW2[0][1] = UPD_MFMA(a0, b1, gW2[0][1]);
                gW2[1][0] = UPD_MFMA(a1, b0, gW2[1][0]);
                gW2[1][1] = UPD_MFMA(a1, b1, gW2[1][1]);
            }
            __builtin_amdgcn_wave_barrier();
            // ---- dh1 = W2^T dz2 (register-chained), dz1 = dh1 (1 - h1^2)
#pragma unroll
            for (int k = 0; k < 32; ++k) {
#pragma unroll
                for (int o = 0; o < 2; ++o) dz1[o] = UPD_MFMA(F2T[k][o][lane], dz2[k >> 4][k & 15], dz1[o]);
            }
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 z = f32x2{dz1[o][r], dz1[o][r + 1]} * one_minus_sq(f32x2{h1[o][r], h1[o][r + 1]});
                    dz1[o][r] = z[0]; dz1[o][r + 1] = z[1];
                }
        } else {
            // The loss derivatives carry 1 / B and whatever the value error is: no fixed scale keeps dz2 inside
            // f16's range.  dz2 is scaled by a power of two (exact) before it is split: 2^(14 - e), e the exponent
            // of the largest |dz2| of a tile.  The dW2 accumulators live in that scale across tiles, so it is kept
            // while a tile's maximum lands in [2^-6, 2^15) (anything smaller still has the absolute accuracy of the
            // tiles that set the scale); when it must change, the accumulators are rescaled (exact; rare: the wave's
            // maximum moved by a binade upwards or by 2^20 downwards).  Never more than 2^60 below the largest tile
            // seen: the accumulators stay far from f32's range.  The epilogue scales them back.
            float mx = 0.0f;
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int r = 0; r < 16; r += 2) mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(dz2[o][r]), __builtin_fabsf(dz2[o][r + 1])));
            const uint32_t mxb = wave_max_bits(__builtin_bit_cast(uint32_t, mx));   // in SGPRs: what follows is scalar code
            if (mxb != 0u) {
                int E = (int)(mxb >> 23);                                       // biased exponent of the wave's maximum
                E = E < 16 ? 16 : (E > 254 ? 254 : E);
                Emax = E > Emax ? E : Emax;
                int En = Ecur;
                if (E > Ecur) En = E;
                else if (E < Ecur - 20) En = E > Emax - 60 ? E : Emax - 60;
                if (En != Ecur) {
#pragma unroll
                    for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                        for (int vt = 0; vt < 2; ++vt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) gW2[ut][vt][r] = __builtin_ldexpf(gW2[ut][vt][r], Ecur - En);
                    Ecur = En;
                }
            }
            const int Eu = Ecur < 16 ? 16 : Ecur;
            const float up = __builtin_bit_cast(float, (uint32_t)(268 - Eu) << 23);     // 2^(14 - e)
            const float down = __builtin_bit_cast(float, (uint32_t)(Eu - 14) << 23);    // 2^(e - 14)
            f32x16 dzs[2];
#pragma unroll
            for (int o = 0; o < 2; ++o) dzs[o] = dz2[o] * up;
            Frag16 bh[4], bl[4];
            split_tile(dzs, bh, bl);
            // ---- dW2 += dz2 h1^T with the samples as k: the pieces go through the staging tiles, two units per word
            put_pieces(0, bh);
            put_pieces(1, bl);
            __builtin_amdgcn_wave_barrier();           // (compiler ordering only: a wave's LDS operations execute in order)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                Frag16 ah[2], al[2], hh[2], hl[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ah[t] = get_pieces(0, t, m); al[t] = get_pieces(1, t, m);
                    hh[t] = get_pieces(2, t, m); hl[t] = get_pieces(3, t, m);
                }
#pragma unroll
                for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                    for (int vt = 0; vt < 2; ++vt) gW2[ut][vt] = mfma16(ah[ut], hl[vt], gW2[ut][vt]);
#pragma unroll
                for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                    for (int vt = 0; vt < 2; ++vt) gW2[ut][vt] = mfma16(al[ut], hh[vt], gW2[ut][vt]);
#pragma unroll
                for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                    for (int vt = 0; vt < 2; ++vt) gW2[ut][vt] = mfma16(ah[ut], hh[vt], gW2[ut][vt]);
            }
            __builtin_amdgcn_wave_barrier();
            // ---- dh1 = W2^T dz2 (register-chained, the same pieces), dz1 = dh1 (1 - h1^2)
            chain16(lds + OFF_F2T, lane, bh, bl, dz1);
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 z = f32x2{dz1[o][r], dz1[o][r + 1]} * (one_minus_sq(f32x2{h1[o][r], h1[o][r + 1]}) * down);
                    dz1[o][r] = z[0]; dz1[o][r + 1] = z[1];
                }
        }

        // ---- dW1 (and db1 in column OBS) += dz1 [x, 1]^T
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) sa[unit(o, r, half)][col] = dz1[o][r];
        if (half == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) sx[k][col] = k < OBS ? x[k] : (k == OBS ? 1.0f : 0.0f);
        }
        __builtin_amdgcn_wave_barrier();
        const int c16 = lane & 15, g16 = lane >> 4;
#pragma unroll 4
        for (int j = 0; j < 8; ++j) {
            const int s = j + 8 * g16;                 // which samples share a k-step is free; this choice halves the bank conflicts
            const float b = c16 < 8 ? sx[c16 & 7][s] : 0.0f;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) gW1[rt] = UPD_MFMA16(sa[16 * rt + c16][s], b, gW1[rt]);
        }

        __builtin_amdgcn_wave_barrier();
        // ---- head weight gradients: rows = head j, columns = h2 units
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) sb[unit(o, r, half)][col] = h2[o][r];
        if (half == 0) {
#pragma unroll
            for (int j = 0; j < NHEAD; ++j) sx[j][col] = dout[j];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int j = 0; j < 8; ++j) {
            const int s = j + 8 * g16;
            const float av = c16 < NHEAD ? sx[c16 & 3][s] : 0.0f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) gWh[ct] = UPD_MFMA16(av, sb[16 * ct + c16][s], gWh[ct]);
        }
    }

    // ---- this wave's partials, in the parameters' own layout, into LDS (every wave is done with the
    // weights and its staging tiles); the four rows are then added in wave order: one row per workgroup
    __syncthreads();
    float *g = lds + wv * L::P;
#pragma unroll
    for (int ut = 0; ut < 2; ++ut)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = unit(ut, r, half);
#pragma unroll
            for (int vt = 0; vt < 2; ++vt) g[L::W2 + u * H + 32 * vt + col] = F16 ? __builtin_ldexpf(gW2[ut][vt][r], Ecur - 141) : gW2[ut][vt][r];
        }
    {
        const int c16 = lane & 15, g16 = lane >> 4;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = 16 * rt + 4 * g16 + r;
                if (c16 < OBS) g[L::W1 + u * OBS + c16] = gW1[rt][r];
                if (c16 == OBS) g[L::B1 + u] = gW1[rt][r];
            }
        // head rows: D row j = register j of lanes 0..15
        if (g16 == 0) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                if (DO_P) {
#pragma unroll
                    for (int j = 0; j < A; ++j) g[L::PW + j * H + 16 * ct + c16] = gWh[ct][j];
                }
                if (DO_V) g[L::CW + 16 * ct + c16] = gWh[ct][3];
            }
        }
    }
    // column sums kept per lane: add the 32 samples (lanes of one half)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = gb2[o][r >> 1][r & 1];
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (col == 0) g[L::B2 + unit(o, r, half)] = v;
        }
#pragma unroll
    for (int j = 0; j < NHEAD; ++j) {
        float v = gbh[j];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) {
            if (DO_P && j < A) g[L::PB + j] = v;
            if (DO_V && j == 3) g[L::CB] = v;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < L::P; k += UPD_BLOCK) {
        float v = lds[k];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += lds[w * L::P + k];
        a.partials[(size_t)blockIdx.x * L::P + k] = v;
    }
#pragma unroll
    for (int k = 0; k < NSCAL; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NSCAL; ++k) a.scal[(size_t)wave_id * NSCAL + k] = acc[k];
    }
}

// grads[p] = sum over the waves' partial rows, f64, in a fixed order (deterministic): a workgroup takes
// 64 consecutive parameters (coalesced row reads), its 16 waves each add every 16th row, and the 16
// sums are added in wave order.  Block 0 also finishes the loss scalars.
constexpr int RED_GROUPS = 16;
__global__ void __launch_bounds__(64 * RED_GROUPS) k_update_reduce(const float *partials, const double *scal, int nrows, int nwaves,
                                                                 int P, long B, float *grads, float *out5, int do_p, int do_v)
{
    __shared__ double part[RED_GROUPS][64];
    const int px = threadIdx.x & 63, ry = threadIdx.x >> 6, p = blockIdx.x * 64 + px;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;          // four independent chains: loads in flight
    if (p < P) {
        int w = ry;
        for (; w + 3 * RED_GROUPS < nrows; w += 4 * RED_GROUPS) {
            s0 += (double)partials[(size_t)w * P + p];
            s1 += (double)partials[(size_t)(w + RED_GROUPS) * P + p];
            s2 += (double)partials[(size_t)(w + 2 * RED_GROUPS) * P + p];
            s3 += (double)partials[(size_t)(w + 3 * RED_GROUPS) * P + p];
        }
        for (; w < nrows; w += RED_GROUPS) s0 += (double)partials[(size_t)w * P + p];
    }
    part[ry][px] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && p < P) {
        double s = 0.0;
        for (int g = 0; g < RED_GROUPS; ++g) s += part[g][px];
        grads[p] = (float)s;
    }
    if (blockIdx.x == 0 && ry < NSCAL) {                      // wave k adds scalar k: a fixed partition, then a butterfly
        double s = 0.0;
        for (int w = px; w < nwaves; w += 64) s += scal[(size_t)w * NSCAL + ry];
        s = wave_sum(s);
        const double n = (double)B;
        const int k = px == 0 ? ry : -1;
        // out5 = {policy_loss, value_loss, clip_frac, kl_div, entropy}; entries the launch does not produce are left alone
        if (k == 0 && do_p) out5[0] = (float)(-s / n);
        if (k == 1 && do_v) out5[1] = (float)(0.5 * s / n);
        if (k == 2 && do_p) out5[2] = (float)(s / n);
        if (k == 3 && do_p) out5[3] = (float)(s / n);
        if (k == 4 && do_p) out5[4] = (float)(s / n);
    }
}

inline int upd_blocks(long B)
{
    const long tiles = (B + 31) / 32, per_block = UPD_BLOCK / 64;
    long blocks = (tiles + per_block - 1) / per_block;
    if (blocks > UPD_MAX_BLOCKS) blocks = UPD_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <int OBS, int A, bool DO_P, bool DO_V>
int launch_update(UpdArgs a, int net_mode, float *grads, float *out5, void *workspace, size_t workspace_bytes, hipStream_t s)
{
    using L = GradLayout<OBS, A, DO_P, DO_V>;
    const int blocks = upd_blocks(a.B), nwaves = blocks * (UPD_BLOCK / 64);
    a.P = L::P;
    const size_t part_bytes = ((size_t)blocks * L::P * sizeof(float) + 255) / 256 * 256;
    PPO_REQUIRE(workspace_bytes >= part_bytes + (size_t)nwaves * NSCAL * sizeof(double));
    a.partials = static_cast<float *>(workspace);
    a.scal = reinterpret_cast<double *>(static_cast<char *>(workspace) + part_bytes);
    if (net_mode == PPO_NET_F16_SPLIT) hipLaunchKernelGGL((k_update<OBS, A, DO_P, DO_V, true>), dim3(blocks), dim3(UPD_BLOCK), 0, s, a);
    else hipLaunchKernelGGL((k_update<OBS, A, DO_P, DO_V, false>), dim3(blocks), dim3(UPD_BLOCK), 0, s, a);
    int rc = ppo_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(k_update_reduce, dim3((L::P + 63) / 64), dim3(64 * RED_GROUPS), 0, s, a.partials, a.scal, blocks, nwaves, L::P, a.B,
                       grads, out5, DO_P ? 1 : 0, DO_V ? 1 : 0);
    return ppo_launch_status();
}

}  // namespace

extern "C" {

size_t ppo_dense_update_workspace_bytes(void)
{
    // the largest gradient row (OBS = 6, A = 3, both heads) for every wave of the largest grid, plus the scalars
    const size_t waves = (size_t)UPD_MAX_BLOCKS * (UPD_BLOCK / 64);
    return ((size_t)UPD_MAX_BLOCKS * GradLayout<6, 3, true, true>::P * sizeof(float) + 255) / 256 * 256 + waves * NSCAL * sizeof(double);
}

size_t ppo_dense_update_grad_floats(int obs_dim, int num_actions, int heads)
{
    if ((obs_dim != 4 && obs_dim != 6) || (num_actions != 2 && num_actions != 3) || heads < 1 || heads > 3) return 0;
    const bool p = heads & 1, v = heads & 2;
    return (size_t)(H * obs_dim + H + H * H + H + (p ? num_actions * H + num_actions : 0) + (v ? H + 1 : 0));
}

int ppo_dense_update_grads(int obs_dim, int num_actions, int heads, int net_mode, const float *w1, const float *b1, const float *w2,
                           const float *b2, const float *policy_w, const float *policy_b, const float *critic_w,
                           const float *critic_b, const long *idx, const float *states, const int *actions, const float *old_lp,
                           const float *old_v, const float *adv, const double *stats3, double clip, int norm_adv,
                           int clip_va, long B, double c_pi, double c_v, double c_ent, float *out5, float *grads,
                           void *workspace, size_t workspace_bytes, void *stream)
{
    PPO_REQUIRE((obs_dim == 4 || obs_dim == 6) && (num_actions == 2 || num_actions == 3));
    PPO_REQUIRE(heads >= 1 && heads <= 3 && B >= 1);
    PPO_REQUIRE(net_mode == PPO_NET_F32 || net_mode == PPO_NET_F16_SPLIT);
    const bool do_p = heads & 1, do_v = heads & 2;
    PPO_REQUIRE(w1 && b1 && w2 && b2 && states && old_v && adv && out5 && grads && workspace);
    PPO_REQUIRE(!do_p || (policy_w && policy_b && actions && old_lp));
    PPO_REQUIRE(!do_v || (critic_w && critic_b));
    PPO_REQUIRE(!norm_adv || stats3);
    PPO_REQUIRE(ppo_aligned(workspace, 8) && ppo_aligned(grads, 4) && ppo_aligned(states, obs_dim == 4 ? 16 : 8));
    UpdArgs a{};
    a.net = Net{w1, b1, w2, b2, policy_w, policy_b, critic_w, critic_b};
    a.idx = idx; a.states = states; a.actions = actions ? actions : reinterpret_cast<const int *>(states);
    a.old_lp = old_lp ? old_lp : old_v; a.old_v = old_v; a.adv = adv; a.stats3 = stats3;
    a.lo = (float)(1.0 - clip); a.hi = (float)(1.0 + clip); a.epsf = (float)clip;
    a.c_pi = (float)c_pi; a.c_v = (float)c_v; a.c_ent = (float)c_ent;
    a.B = B; a.norm_adv = norm_adv; a.clip_va = clip_va;
    hipStream_t s = (hipStream_t)stream;
#define UPD_CASE(OBS_, A_)                                                                                             \
    if (obs_dim == OBS_ && num_actions == A_) {                                                                        \
        if (do_p && do_v) return launch_update<OBS_, A_, true, true>(a, net_mode, grads, out5, workspace, workspace_bytes, s);   \
        if (do_p) return launch_update<OBS_, A_, true, false>(a, net_mode, grads, out5, workspace, workspace_bytes, s);          \
        return launch_update<OBS_, A_, false, true>(a, net_mode, grads, out5, workspace, workspace_bytes, s);                    \
    }
    UPD_CASE(4, 2) UPD_CASE(4, 3) UPD_CASE(6, 2) UPD_CASE(6, 3)
#undef UPD_CASE
    return (int)hipErrorInvalidValue;
}

}  // extern "C"
