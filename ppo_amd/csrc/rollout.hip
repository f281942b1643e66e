// Whole-rollout kernel: T time steps of policy forward + action sampling + env transition in ONE
// launch -- gfx950.
//
// Replaces the T-step loop of SyncVecEnv.rollout (reference src/ppo.py:320-398) for the dense
// 64-64 tanh network with a discrete head (Agent.get_actions_and_values, src/ppo.py:166-216,
// network of :124-145) on CartPole / Acrobot.  Envs are independent and the policy is frozen for
// the whole rollout, so nothing couples lanes: each lane owns one env for all T steps and no
// inter-lane or inter-launch synchronisation exists.  The per-step-launch path (torch MLP + env.hip)
// spends ~85 us of kernel time and ~150 us of wall per time step at N = 65536, almost all of it
// in launch gaps and in streaming 16 MB hidden activations through HBM for a 0.6 GFLOP problem.
// Here the activations never leave the register file.
//
// Per wave (64 envs, or 32 for small batches) and step the two 64-wide tanh layers run on the matrix
// cores, in one of two forms (net_mode): v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate -- exactly a
// k-ordered fmaf chain), or, by default, v_mfma_f32_32x32x16_f16 on two-piece f16 splits of both
// operands (further down).  In both the layers are CHAINED IN REGISTERS: the D layout of one 32x32
// tile (lane = env l & 31, register r = unit (r&3) + 8 (r>>2) + 4 (l>>5)) is, register by
// register, a valid B operand of the next layer for a k order the host is free to choose -- so the
// next layer's weights are packed on the host in THAT k order and an accumulator register goes
// straight from tanh into the next MFMA.  No LDS round trip, no shuffles.  The backbone's A-operand
// fragments stay in the 512-entry register file for all T steps (one wave per SIMD); the f16 form
// reads the head weights (and a split net's value backbone) from LDS.  Earlier formulations,
// measured at N = 65536, T = 128, joint net:
//   - weights as SGPR operands via the scalar cache (VALU dot products)          5.4 ms / rollout
//   - weights broadcast from LDS with ds_read_b128 (VALU)                         2.9 ms
//   - MFMA with weights and a [unit][env] activation tile in LDS                  1.56 ms
//   - f32 MFMA, register-chained (net_mode 0 today)                               1.07 ms
//   - f16-split MFMA, register-chained (net_mode 1 today)                         0.56 ms
// The heads are per-lane partial dot products over the 32 units a lane holds, completed with one
// v_permlane32_swap + add, which also hands every lane the result of ITS env (lane = env again for
// sampling (Philox) and the fp64 env physics of env_core.h).
//
// Bound: VALU issue (f16 form: 1243 VALU instructions per wave and step, 70 % of the wave's cycles;
// the f32 form is bound by its 144 x 64 MFMA cycles, which do not overlap with the wave's VALU
// work).  HBM: 40 B written per env-step (states 16, action, log-prob, value, reward, done, trunc).
#include <type_traits>
#include "env_core.h"
#include "net_layout.h"
#include "mfma_split.h"
#ifdef PPO_ROLLOUT_ABLATE_BUILD
#include <cstdlib>
#endif

using namespace ppo_env;

// The ablation switches (PPO_ROLLOUT_ABLATE) are compiled in only on request: as run-time branches
// they cut the step into basic blocks and keep the scheduler from interleaving across them.
#ifdef PPO_ROLLOUT_ABLATE_BUILD
#define ABLATE(args) ((args).ablate)
#else
#define ABLATE(args) 0
#endif

namespace {

using namespace ppo_net;   // H, ksteps, backbone_floats, HEAD_FLOATS, BB16_* (net_layout.h)

struct RolloutArgs {
    EnvArgs env;             // state, counters, rings, reset stream (action/obs_out/... unused)
    const float *params;     // packed network, layout below
    uint64_t sample_step0;   // Philox position of time step 0 of this rollout
    float *obs_io;           // [N,OBS] current observations in; carried observations out
    float *states;           // [T,N,OBS]
    int *actions;            // [T,N]
    float *log_probs, *values, *rewards;  // [T,N]
    int *done, *trunc;       // [T,N]
    uint32_t *flag_words;    // [ceil(T/16),N] or NULL: bits 2k / 2k+1 of word q = done / trunc of row 16 q + k (ppo_gae_dense_packed)
    float *term_obs;         // [T,N,OBS], written where a flag is raised
    float *ev_dense;         // [T,N] or NULL: V(terminal obs) written where trunc != 0
    double *reward_sum;      // or NULL: += sum of all rewards of this rollout (one f64 atomic per wave)
    int T;
    int ablate;              // diagnostic only (PPO_ROLLOUT_ABLATE): 1 skip network, 2 skip physics, 4 skip sampling
};

// Packed parameter layout (floats), built by ppo_amd.ops.pack_dense_agent:
//   backbone(pi): F1[S1][2][64]  F2[33][2][64]      MFMA A-operand fragments
//   backbone(va): same (split nets only)
//   heads:        HL[4][2][16][64]  per-lane head weights, outputs j = 0..2 logits (zero padded),
//                 j = 3 critic;  bias4[4] = policy bias (zero padded), critic bias
// unit(o, r, h) = 32 o + (r & 3) + 8 (r >> 2) + 4 h is the hidden unit that register r of output
// tile o holds in lane half h (the MFMA D layout).  A Linear(K -> 64) layer y = W x + b is 1 + K/2
// k-steps per 32-output tile o; lane l of k-step s holds
//   s = 0           : (l < 32) ? b[32 o + l] : 0                   (paired with the B row [1; 0])
//   layer 1, s > 0  : W[32 o + (l & 31)][2 (s - 1) + (l >> 5)]     (inputs in natural order)
//   layer 2, s > 0  : W[32 o + (l & 31)][unit(o_in, r_in, l >> 5)],  s = 1 + 16 o_in + r_in
//   HL[j][o][r][l]  : Whead_j[unit(o, r, l >> 5)]
// (ksteps, backbone_floats, HEAD_FLOATS, BB16_*: net_layout.h, shared with the pack kernel in pack.hip)


// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp units: 4 VALU instructions, no
// branches (ocml's tanhf is two divergent branches of ~20).  exp2 overflow gives rcp(inf) = 0 ->
// exactly 1, underflow gives rcp(1) -> exactly -1.  Absolute error < 2e-7, i.e. the size of one
// f32 rounding of an O(1) activation.
__device__ __forceinline__ float fast_tanh(float x)
{
    const float t = __builtin_amdgcn_exp2f(2.8853900817779268f * x);   // exp(2x)
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);   // 2 r is exact: same bits as 1 - 2 r
}

// lanes 32..63 of a <-> lanes 0..31 of b.  (The compiler builtin for this instruction returns its
// second result wrong in ROCm 7.2's hipcc, hence the asm; the s_nops cover the VALU-write ->
// permlane-read wait states the assembler does not insert for inline asm.)
__device__ __forceinline__ void swap_halves(float &a, float &b)
{
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

// A-operand fragments of one backbone held in registers (all indices are compile-time constants
// after unrolling) or read from LDS (pointer already offset by the lane).
template <int S> struct RegFrags {
    float a[S][2];
    __device__ __forceinline__ void load(const float *g, int lane)
    {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            a[s][0] = g[(s * 2 + 0) * 64 + lane];
            a[s][1] = g[(s * 2 + 1) * 64 + lane];
        }
    }
    __device__ __forceinline__ float get(int s, int o) const { return a[s][o]; }
};
struct LdsFrags {
    const float *p;
    __device__ __forceinline__ float get(int s, int o) const { return p[(s * 2 + o) * 64]; }
};

__device__ __forceinline__ f32x16 zero16()
{
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}

#define PPO_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32((a_), (b_), (c_), 0, 0, 0)

// Both tanh layers and the heads for the 64 envs of this wave.  obs: this lane's env (lane = env).
// out[j]: head j of this lane's env, bias not yet added (j < A logits if DO_P, j = 3 critic if DO_V).
template <int OBS, int A, bool DO_P, bool DO_V, class W1, class W2>
__device__ __forceinline__ void net_forward(const float (&obs)[6], const W1 &w1, const W2 &w2,
                                            const float (&hl)[4][2][16], int half, float (&out)[4])
{
    constexpr int S1 = ksteps(OBS);
    const float one = half == 0 ? 1.0f : 0.0f;
    // ---- layer 1: bias k-step, then input pairs; one half-swap turns (obs[2k], obs[2k+1]) of the
    // 64 envs into the B fragments of env tile 0 and env tile 1.
    f32x16 h1[2][2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int e = 0; e < 2; ++e) h1[o][e] = PPO_MFMA(w1.get(0, o), one, zero16());
#pragma unroll
    for (int s = 1; s < S1; ++s) {
        float b0 = obs[2 * (s - 1)], b1 = obs[2 * (s - 1) + 1];
        swap_halves(b0, b1);
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            h1[o][0] = PPO_MFMA(w1.get(s, o), b0, h1[o][0]);
            h1[o][1] = PPO_MFMA(w1.get(s, o), b1, h1[o][1]);
        }
    }
    // ---- layer 2 as four 33-MFMA passes, one 32 x 32 output tile (op, e) each.  tanh(h1) registers
    // are the B operands as they are.  The VALU work is software-pipelined into the MFMA issue
    // shadows by hand (one sched_barrier-delimited slot per MFMA; left alone, the scheduler emits a
    // pass's MFMAs back to back and the tanh / head work after them):
    //   pass (0,0): tanh of the h1 registers of env tile 0, one k-step ahead of their use
    //   pass (1,0): heads of tile (0,0)        pass (0,1): tanh of h1 env tile 1 + heads of (1,0)
    //   pass (1,1): heads of tile (0,1)        tail:       heads of tile (1,1)
    float p[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j][0] = p[j][1] = 0.0f;
    f32x16 h2[2][2];
    auto head = [&](int op, int e, int r) {
        const float h = fast_tanh(h2[op][e][r]);
        if (DO_P) {
#pragma unroll
            for (int j = 0; j < A; ++j) p[j][e] = __builtin_fmaf(h, hl[j][op][r], p[j][e]);
        }
        if (DO_V) p[3][e] = __builtin_fmaf(h, hl[3][op][r], p[3][e]);
    };
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        h1[0][e][0] = fast_tanh(h1[0][e][0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int op = 0; op < 2; ++op) {
            h2[op][e] = PPO_MFMA(w2.get(0, op), one, zero16());
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int o = k >> 4, r = k & 15;
                h2[op][e] = PPO_MFMA(w2.get(1 + k, op), h1[o][e][r], h2[op][e]);
                if (op == 0 && k + 1 < 32) h1[(k + 1) >> 4][e][(k + 1) & 15] = fast_tanh(h1[(k + 1) >> 4][e][(k + 1) & 15]);
                if ((k & 1) == 0) {                    // heads of the tile finished one pass ago
                    if (op == 1) head(0, e, k >> 1);
                    else if (e == 1) head(1, 0, k >> 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) head(1, 1, r);
    // ---- complete the dot products across the two lane halves; lane l ends with env l's result
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if ((j < A && DO_P) || (j == 3 && DO_V)) {
            swap_halves(p[j][0], p[j][1]);
            out[j] = p[j][0] + p[j][1];
        }
    }
}

constexpr int ROLLOUT_BLOCK = 256;  // 4 waves, one per SIMD

// ---- network mode 0: f32 MFMA, every product an exact f32 FMA --------------------------------
template <int OBS_, int A_, bool JOINT_> struct NetF32 {
    static constexpr int OBS = OBS_, A = A_, WAVE_ENVS = 64;
    static constexpr bool JOINT = JOINT_, DIFF = false;
    static constexpr bool FULL_OK = false;         // (k_rollout's FULL form: the yardstick mode keeps its one form)
    static constexpr int NB = backbone_floats(OBS);
    static constexpr int S1 = ksteps(OBS), S2 = ksteps(H);
    static constexpr int LDS_WORDS = JOINT ? 1 : NB;
    RegFrags<S1> P1;
    RegFrags<S2> P2;
    LdsFrags V1, V2;
    float hl[4][2][16];
    float bias4[4];
    int half;

    // policy (or joint) backbone in registers for the whole rollout; the value backbone of split
    // nets is shared by the block's waves through LDS.
    __device__ __forceinline__ void load(const float *__restrict__ params, float *smem, int lane)
    {
        half = lane >> 5;
        P1.load(params, lane);
        P2.load(params + S1 * 128, lane);
        if (!JOINT) {
            for (int k = threadIdx.x; k < NB; k += ROLLOUT_BLOCK) smem[k] = params[NB + k];
            __syncthreads();                       // the only barrier: waves are independent from here on
        }
        V1.p = smem + lane;
        V2.p = smem + (JOINT ? 0 : S1 * 128) + lane;
        const float *heads = params + NB * (JOINT ? 1 : 2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    hl[j][o][r] = (j < A || j == 3) ? heads[((j * 2 + o) * 16 + r) * 64 + lane] : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) bias4[j] = heads[4 * 2 * 16 * 64 + j];
    }
    // VNET: use the value backbone (split nets)
    template <bool DO_P, bool DO_V, bool VNET>
    __device__ __forceinline__ void forward(const float (&obs)[6], float (&out)[4]) const
    {
        if (VNET) net_forward<OBS, A, DO_P, DO_V>(obs, V1, V2, hl, half, out);
        else net_forward<OBS, A, DO_P, DO_V>(obs, P1, P2, hl, half, out);
    }
};

// ---- network mode 1: f16 MFMA on two-piece splits (f32-grade products at 5x the MFMA rate) -----
// v_mfma_f32_32x32x2_f32 costs 64 cycles per 2 k and does not overlap with the wave's other VALU
// work (measured: SQ_VALU_MFMA_COEXEC_CYCLES = 0, issue gaps add up); v_mfma_f32_32x32x16_f16 does
// 16 k in 32 cycles.  Every operand x is split as x = xh + xl with xh = f16(x) (to nearest) and
// xl = f16(x - xh) (the subtraction is exact), 22 significant bits in all -- xl may be an f16
// subnormal, which the matrix cores honour: absolute error <= 2^-25 -- and a product becomes
// xh wl + xl wh + xh wh: three MFMAs into one f32 accumulator (small terms first), the dropped
// xl wl term being below 2^-22 relative.  f16 x f16 products are exact in the f32 accumulate.
// The 2 log2(e) factor of tanh's exp2 is folded into the packed weights and biases.
// Layouts (host: ppo_amd.ops.pack_dense_agent(mode=1)), per backbone, in 32-bit words:
//   W1[op 2][piece 2][lane 64][4]   8 f16 per lane: k = i (both lane halves): W1[32 op + (l&31)][i], i < OBS;
//                                   b1[32 op + (l&31)] at i = OBS (the input vector carries a 1.0 there)
//   W2[j 4][op 2][piece 2][lane 64][4]   k-step j = 2 o_in + (r_in >> 3), element i = r_in & 7:
//                                   W2[32 op + (l&31)][unit(o_in, r_in, l >> 5)]
//   B2[op 2][r 16][lane 64] f32     b2[unit(op, r, l >> 5)]  (the C operand of the first k-step)
// tanh of an argument already multiplied by 2 log2(e)
__device__ __forceinline__ float tanh_prescaled(float x2)
{
    // one fma: 2 r is exact, so this is bit-identical to 1 - 2 r and one instruction shorter
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x2) + 1.0f), 1.0f);
}

// two at a time: the add and the fma as packed f32 instructions (one issue slot for two values)
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ void tanh2_prescaled(float x0, float x1, float &a0, float &a1)
{
    f32x2 e = {__builtin_amdgcn_exp2f(x0), __builtin_amdgcn_exp2f(x1)};
    e += 1.0f;
    const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    const f32x2 t = __builtin_elementwise_fma(f32x2{-2.0f, -2.0f}, r, f32x2{1.0f, 1.0f});
    a0 = t[0];
    a1 = t[1];
}
__device__ __forceinline__ f32x2 tanh2v_prescaled(float x0, float x1)
{
    f32x2 e = {__builtin_amdgcn_exp2f(x0), __builtin_amdgcn_exp2f(x1)};
    e += 1.0f;
    const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    return __builtin_elementwise_fma(f32x2{-2.0f, -2.0f}, r, f32x2{1.0f, 1.0f});
}

// ---- network mode 2: bf16 MFMA on three-piece splits (every f32 operand represented exactly) ----
// x = x0 + x1 + x2 with bf16 pieces (8 significand bits each, f32's exponent range; mfma_split.h), so nothing
// is lost in the operands -- weights or activations, large or tiny.  A product x w is the six cross terms of
// order <= 2 (x0 w0; x0 w1, x1 w0; x0 w2, x1 w1, x2 w0) as v_mfma_f32_32x32x16_bf16 into ONE f32 accumulator,
// small terms first; the three dropped terms (x1 w2, x2 w1, x2 w2) are below 2^-23 |x w| in the worst case
// and ~2^-25 on average, the size of ONE f32 rounding -- and the matrix core adds its sixteen exact products
// of a k-step with fewer roundings than a sixteen-step fmaf chain.  Measured (tests/test_gpu_net_precision.py,
// profiles/r03_net_precision.txt): as accurate as the exact-f32 kernels at every weight scale.  192 MFMA
// cycles per 16 k against 512 for v_mfma_f32_32x32x2_f32, and bf16 MFMAs co-execute with the wave's VALU work.
// Layouts as in mode 1 with three pieces instead of two (W1[op][piece 3][lane][4], W2[j][op][piece 3][lane][4]).
struct SchemeF16 {          // mode 1: two f16 pieces, three terms
    static constexpr int P = 2, NT = 3;
    struct Ctx { __device__ __forceinline__ void init() {} };
    __device__ static constexpr int tw(int t) { return t == 0 ? 1 : 0; }      // weight piece of term t
    __device__ static constexpr int ta(int t) { return t == 1 ? 1 : 0; }      // activation piece of term t
    __device__ static __forceinline__ f32x16 mfma(const Frag16 &a, const Frag16 &b, f32x16 c) { return mfma16(a, b, c); }
    __device__ static __forceinline__ void split(float a0, float a1, const Ctx &, uint32_t (&p)[2]) { split2(a0, a1, p[0], p[1]); }
};
struct SchemeBF16 {         // mode 2: three bf16 pieces, six terms, ascending magnitude
    static constexpr int P = 3, NT = 6;
    using Ctx = Bf16Select;
    __device__ static constexpr int tw(int t) { return t == 0 ? 2 : (t == 1 || t == 3) ? 1 : 0; }   // w2 a0, w1 a1, w0 a2, w1 a0, w0 a1, w0 a0
    __device__ static constexpr int ta(int t) { return t == 2 ? 2 : (t == 1 || t == 4) ? 1 : 0; }
    __device__ static __forceinline__ f32x16 mfma(const Frag16 &a, const Frag16 &b, f32x16 c) { return mfma16_bf16(a, b, c); }
    __device__ static __forceinline__ void split(float a0, float a1, const Ctx &c, uint32_t (&p)[3]) { split3_bf16(a0, a1, c, p[0], p[1], p[2]); }
};

template <int P> struct RegW16 {
    Frag16 w1[2][P], w2[4][2][P];
    f32x16 b2[2];
    __device__ __forceinline__ f32x16 bias2(int op) const { return b2[op]; }
    __device__ __forceinline__ void load(const float *__restrict__ g, int lane)
    {
        const uint4 *v = reinterpret_cast<const uint4 *>(g);
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
            for (int pc = 0; pc < P; ++pc) {
                const uint4 x = v[(op * P + pc) * 64 + lane];
                w1[op][pc] = Frag16{{x.x, x.y, x.z, x.w}};
            }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int op = 0; op < 2; ++op)
#pragma unroll
                for (int pc = 0; pc < P; ++pc) {
                    const uint4 x = v[bb16_w1(P) / 4 + ((j * 2 + op) * P + pc) * 64 + lane];
                    w2[j][op][pc] = Frag16{{x.x, x.y, x.z, x.w}};
                }
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
            for (int r = 0; r < 16; ++r) b2[op][r] = g[bb16_w1(P) + bb16_w2(P) + (op * 16 + r) * 64 + lane];
    }
    __device__ __forceinline__ Frag16 get1(int op, int pc) const { return w1[op][pc]; }
    __device__ __forceinline__ Frag16 get2(int j, int op, int pc) const { return w2[j][op][pc]; }
};
template <int P> struct LdsW16 {
    const float *p;   // LDS copy of one backbone
    int lane;
    __device__ __forceinline__ Frag16 rd(int idx) const
    {
        const uint4 x = reinterpret_cast<const uint4 *>(p)[idx * 64 + lane];
        return Frag16{{x.x, x.y, x.z, x.w}};
    }
    __device__ __forceinline__ Frag16 get1(int op, int pc) const { return rd(op * P + pc); }
    __device__ __forceinline__ Frag16 get2(int j, int op, int pc) const { return rd(bb16_w1(P) / 256 + (j * 2 + op) * P + pc); }
    __device__ __forceinline__ f32x16 bias2(int op) const
    {
        f32x16 b;
#pragma unroll
        for (int r = 0; r < 16; ++r) b[r] = p[bb16_w1(P) + bb16_w2(P) + (op * 16 + r) * 64 + lane];
        return b;
    }
};

// Per-lane head weights HL[j][op][r], read from an LDS copy at use (the LDS pipe is otherwise idle
// in this kernel and the registers are needed elsewhere).
struct LdsHL {
    const float *p;   // LDS copy of HL as [j][op][r / 4][lane][r % 4], already offset by 4 lane:
                      // four consecutive r are one conflict-free ds_read_b128
    __device__ __forceinline__ float4 get4(int j, int op, int r4) const
    {
        return *reinterpret_cast<const float4 *>(p + ((j * 2 + op) * 4 + r4) * 256);
    }
};

// ETILES = 2: the wave's 64 lanes are 64 envs, two 32-column MFMA tiles.  ETILES = 1: lanes 0-31 are
// 32 envs, one tile, lanes 32-63 only hold the other halves of its MFMA operands (their outputs are
// meaningless): half the network work per wave, for batches too small to give every SIMD a wave of 64.
// NP: policy rows in the head table.  A two-action policy keeps ONE row, the difference of its two
// logit rows (NetF16::load): softmax and sampling only depend on l1 - l0.
template <class S, int OBS, int NP, bool DO_P, bool DO_V, int ETILES, class W, class HLT>
__device__ __forceinline__ void net_forward16(const float (&obs)[6], const W &w, const HLT &hl, int half,
                                              const typename S::Ctx &ctx, float (&out)[4])
{
    constexpr int A = NP, P = S::P;
    static_assert(OBS < 8, "layer-1 inputs and the bias slot must fit one 8-wide k half");
    // ---- input pieces: (obs, 1.0 for the bias, zeros); env tile e reads them from lane half e
    float in[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) in[k] = k < OBS ? obs[k] : (k == OBS ? 1.0f : 0.0f);
    Frag16 ip[P];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t pc[P];
        S::split(in[2 * i], in[2 * i + 1], ctx, pc);
#pragma unroll
        for (int q = 0; q < P; ++q) ip[q].w[i] = pc[q];
    }
    Frag16 b2[P][ETILES][4];                   // layer-2 B operands: pieces of tanh(h1), [piece][env tile][k-step]
#pragma unroll
    for (int e = 0; e < ETILES; ++e) {
        Frag16 bp[P];
#pragma unroll
        for (int q = 0; q < P; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) bp[q].w[i] = half == e ? ip[q].w[i] : 0u;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            f32x16 m = zero16();                               // small cross terms first
#pragma unroll
            for (int t = 0; t < S::NT; ++t) m = S::mfma(w.get1(o, S::tw(t)), bp[S::ta(t)], m);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float a0, a1;
                tanh2_prescaled(m[r], m[r + 1], a0, a1);
                uint32_t pc[P];
                S::split(a0, a1, ctx, pc);
#pragma unroll
                for (int q = 0; q < P; ++q) b2[q][e][2 * o + (r >> 3)].w[(r & 7) >> 1] = pc[q];
            }
        }
    }
    // ---- layer 2 + heads, one 32 x 32 output tile at a time.  Head dot products accumulate two
    // hidden units at a time (v_pk_fma_f32: tanh hands its outputs over in register pairs, the head
    // weights of consecutive units are adjacent in the ds_read_b128 result), even and odd units in
    // the two halves of the accumulator, added once at the end.
    f32x2 p2[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) p2[j][0] = p2[j][1] = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int e = 0; e < ETILES; ++e) {
        // first group of head weights: requested before the MFMA chain, in registers when it ends
        float4 wn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((j < A && DO_P) || (j == 3 && DO_V)) wn[j] = hl.get4(j, 0, 0);
        // the two output tiles' accumulator chains interleave: each MFMA depends on the one before last
        f32x16 m[2] = {w.bias2(0), w.bias2(1)};
        if constexpr (P == 2) {                    // (this interleaving of the two small terms is the measured one)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int op = 0; op < 2; ++op) m[op] = S::mfma(w.get2(j, op, 1), b2[0][e][j], m[op]);
#pragma unroll
                for (int op = 0; op < 2; ++op) m[op] = S::mfma(w.get2(j, op, 0), b2[1][e][j], m[op]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int op = 0; op < 2; ++op) m[op] = S::mfma(w.get2(j, op, 0), b2[0][e][j], m[op]);
        } else {
#pragma unroll
            for (int t = 0; t < S::NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int op = 0; op < 2; ++op) m[op] = S::mfma(w.get2(j, op, S::tw(t)), b2[S::ta(t)][e][j], m[op]);
        }
        // heads: four units at a time; the weights of the next group are requested from LDS before
        // the current group is consumed, so that the read latency hides behind its tanh work
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int op = g >> 2, r0 = 4 * (g & 3);
            float4 wc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((j < A && DO_P) || (j == 3 && DO_V)) {
                    wc[j] = wn[j];
                    if (g + 1 < 8) wn[j] = hl.get4(j, (g + 1) >> 2, (g + 1) & 3);
                }
            const f32x2 h01 = tanh2v_prescaled(m[op][r0], m[op][r0 + 1]);
            const f32x2 h23 = tanh2v_prescaled(m[op][r0 + 2], m[op][r0 + 3]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((j < A && DO_P) || (j == 3 && DO_V)) {
                    p2[j][e] = __builtin_elementwise_fma(h01, f32x2{wc[j].x, wc[j].y}, p2[j][e]);
                    p2[j][e] = __builtin_elementwise_fma(h23, f32x2{wc[j].z, wc[j].w}, p2[j][e]);
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if ((j < A && DO_P) || (j == 3 && DO_V)) {
            float p0 = p2[j][0][0] + p2[j][0][1], p1 = ETILES == 2 ? p2[j][1][0] + p2[j][1][1] : 0.0f;
            swap_halves(p0, p1);
            out[j] = p0 + p1;
        }
    }
}

template <class S, int OBS_, int A_, bool JOINT_, int ETILES_> struct NetF16 {
    static constexpr int OBS = OBS_, A = A_, WAVE_ENVS = 32 * ETILES_;
    static constexpr int BB16_WORDS = bb16_words(S::P), BB16_B2 = ppo_net::BB16_B2;
    static constexpr bool JOINT = JOINT_;
    // k_rollout's FULL form (one scheduling region per step) only for the two-piece scheme, where it gains 2.4 %; with
    // three pieces it gains 0.3 %, not worth a third instantiation per env.  (The first FULL build of the three-piece
    // split-net CartPole kernel also returned NaN log-probs in a few waves per launch: hipcc 7.2 lost the fourth dword of
    // a spilled weight fragment -- tools/check_spills.py, run by the Makefile on every build since, and DESIGN.md 4.2.)
#ifdef PPO_ROLLOUT_FULL_ALL                        // (development: tools/build_variant.sh x -DPPO_ROLLOUT_FULL_ALL)
    static constexpr bool FULL_OK = ETILES_ == 2;
#else
    static constexpr bool FULL_OK = S::P == 2 && ETILES_ == 2;
#endif
    // Two actions: only l1 - l0 matters to softmax, log-probs and the sampled action, so the head
    // table holds the difference row (and bias) in slot 0 -- one policy dot product instead of two;
    // forward() then returns out[0] = l1 - l0 (bias4[0] = b1 - b0).
    static constexpr bool DIFF = A_ == 2;
    static constexpr int NP = DIFF ? 1 : A_;
    static constexpr int HL_WORDS = 4 * 2 * 16 * 64;
    static constexpr int LDS_WORDS = (JOINT ? 0 : BB16_WORDS) + HL_WORDS + BB16_B2 + 4;
    RegW16<S::P> P;
    LdsW16<S::P> V;
    LdsHL hs;
    typename S::Ctx ctx;
    float bias4[4];
    int half;

    // Head weights live in LDS for joint and split nets alike: with the MFMA results in VGPRs
    // everything the VALU touches has to fit 256 registers (measured, CartPole joint: 0.68 ms with
    // the head weights in registers, 0.65 ms from LDS); split nets keep the value backbone there too.
    __device__ __forceinline__ void load(const float *__restrict__ params, float *smem, int lane)
    {
        half = lane >> 5;
        ctx.init();
        P.load(params, lane);
        const float *heads = params + BB16_WORDS * (JOINT ? 1 : 2);
        float *shl = smem + (JOINT ? 0 : BB16_WORDS);
        if (!JOINT) {
            for (int k = threadIdx.x; k < BB16_WORDS; k += ROLLOUT_BLOCK) smem[k] = params[BB16_WORDS + k];
        }
        // k = (jo * 16 + r) * 64 + lane; thread tid takes k = tid + 256 i, so jo = i / 4 is a compile-time constant of
        // the unrolled loop: no branch inside it, every load of the table issued before the first is waited for.  (As a
        // run-time loop with the DIFF test inside, hipcc made it 32 trips of load - wait - store: 8.5 us at the head
        // of every launch, tools/rollout_timeline.py.)  Rows no forward pass reads are not copied.
        static_assert(HL_WORDS % ROLLOUT_BLOCK == 0 && ROLLOUT_BLOCK == 256, "head-table fill: four trips per (j, op) row");
        float hv[HL_WORDS / ROLLOUT_BLOCK];
#pragma unroll
        for (int i = 0; i < HL_WORDS / ROLLOUT_BLOCK; ++i) {
            const int k = (int)threadIdx.x + i * ROLLOUT_BLOCK, j = i >> 3;
            hv[i] = 0.0f;
            if (j < NP || j == 3) {
                hv[i] = heads[k];
                if (DIFF && j == 0) hv[i] = heads[k + 2 * 16 * 64] - hv[i];   // row 0 <- row 1 - row 0
            }
        }
#pragma unroll
        for (int i = 0; i < HL_WORDS / ROLLOUT_BLOCK; ++i) {
            const int k = (int)threadIdx.x + i * ROLLOUT_BLOCK, j = i >> 3;
            const int l = k & 63, r = (k >> 6) & 15, jo = k >> 10;
            if (j < NP || j == 3) shl[((jo * 4 + (r >> 2)) * 64 + l) * 4 + (r & 3)] = hv[i];
        }
        __syncthreads();                           // the only barrier: waves are independent from here on
        V.p = smem;
        V.lane = lane;
        hs.p = shl + 4 * lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) bias4[j] = heads[4 * 2 * 16 * 64 + j];
        if (DIFF) bias4[0] = bias4[1] - bias4[0];
    }
    template <bool DO_P, bool DO_V, bool VNET>
    __device__ __forceinline__ void forward(const float (&obs)[6], float (&out)[4]) const
    {
        if (VNET) net_forward16<S, OBS, NP, DO_P, DO_V, ETILES_>(obs, V, hs, half, ctx, out);
        else net_forward16<S, OBS, NP, DO_P, DO_V, ETILES_>(obs, P, hs, half, ctx, out);
    }
};

// FULL: every lane of every wave of the grid owns an env (64-env waves, N a multiple of the 256 envs of a workgroup -- the
// large batches this kernel is for).
// `live` is then a compile-time true, and with it goes the exec-masked region around the env transition and the
// stores: the step becomes ONE scheduling region, in which hipcc moves the part of the transition that does not depend
// on the action (the pole angle's sin / cos, the products of the old state) up beside the network's MFMA chains
// instead of starting it after the last head FMA.  A/B at C2, alternating fresh processes: 427.5 -> 417.2 us per
// rollout (f16-split), 601.6 -> 599.6 us (bf16x3).  Same instructions per lane, bit-identical results.
#ifdef PPO_ROLLOUT_TIMELINE                        // development (tools/rollout_timeline.py): where a launch's fixed part goes
__device__ unsigned long long g_rollout_timeline[4096 * 8];
#define TL_MARK(k) do { if ((threadIdx.x & 63) == 0) { const size_t w_ = ((size_t)blockIdx.x * ROLLOUT_BLOCK + threadIdx.x) >> 6; \
    if (w_ < 4096) g_rollout_timeline[w_ * 8 + (k)] = wall_clock64(); } } while (0)
#else
#define TL_MARK(k) do { } while (0)
#endif
template <class Env, class Net, bool DRAWS, bool FULL = false>
__global__ void __launch_bounds__(ROLLOUT_BLOCK) k_rollout(RolloutArgs a, const float *__restrict__ params)
{
    TL_MARK(0);
    constexpr int OBS = Env::OBS, A = Net::A;
    constexpr bool JOINT = Net::JOINT;
    static_assert(OBS % 2 == 0 && A <= 3, "fragment layout assumes an even input width and <= 3 actions");

    constexpr int WE = Net::WAVE_ENVS;             // envs per wave: 64, or 32 in lanes 0-31
    // 32-env waves: lane l + 32 carries a copy of lane l's env anyway (MFMA needs all 64 lanes); for an env whose
    // transition is dominated by pairs of trigonometric reductions (Acrobot) it takes one of each pair
    // (env_core.h: Acrobot::trig_of<PAIR>) instead of idling through the physics
    constexpr bool PAIR = WE == 32 && Env::PAIR_TRIG;
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * ROLLOUT_BLOCK + threadIdx.x) >> 6;
    const size_t n = wave * WE + (WE == 64 ? lane : (lane & 31));
    // Lanes without an env of their own (n >= N, or the upper half of a 32-env wave) stay: MFMA needs
    // all 64 lanes.  They shadow a valid env and never store.
    static_assert(!FULL || WE == 64, "FULL: 64-env waves");
    const bool live = FULL || (n < (size_t)a.env.N && (WE == 64 || lane < 32));
    const size_t nn = n < (size_t)a.env.N ? n : (size_t)a.env.N - 1;

    __shared__ __attribute__((aligned(16))) float smem[Net::LDS_WORDS];
    Net net;
    net.load(params, smem, lane);
    TL_MARK(1);

    EnvRegs e;
    load_env(e, a.env, nn);
    typename Env::Aux aux;                         // what the env derives from its state and carries along with it
    // an env with TWO_FORMS of its transition (Acrobot, env_core.h): a wave that loaded a state no transition leaves
    // behind runs its FIRST step in the GENERAL form, ahead of the loop; the loop itself only holds the fast form
    const bool general_form = Env::TWO_FORMS && __builtin_amdgcn_readfirstlane((int)Env::wide(e.s)) != 0;   // (uniform, and known to be)
    if (general_form) Env::template prime<PAIR, true>(e.s, aux);
    else Env::template prime<PAIR>(e.s, aux);
    ResetAhead ahead;
    float obs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < OBS; ++k) obs[k] = a.obs_io[nn * OBS + k];

    const size_t N = (size_t)a.env.N;
    float rsum = 0.0f;
    // Everything loaded above has arrived before the loop starts, so that no vmcnt wait is needed
    // inside it: there it would also wait for the previous step's stores (vmcnt counts both).
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0), other counters untouched
    TL_MARK(2);
    int pend_t = -1;                               // row of this env's truncation still waiting for its bootstrap value
    // Second network pass over the terminal observations of the envs with a pending truncation.  always_inline: given
    // a third call site hipcc made it a real function, reached by s_swappc with the network's registers handed over
    // through scratch -- and its upper half-wave came back with f16-grade values (tests/test_gpu_bench_shapes.py,
    // Acrobot, while this kernel held two step loops); tools/check_spills.py now refuses calls inside a kernel
    auto settle = [&]() __attribute__((always_inline)) {
        const bool has = pend_t >= 0;
        float tobs[6];
        __builtin_amdgcn_s_waitcnt(0x0F70);        // this wave's term_obs stores have landed (rare path)
#pragma unroll
        for (int k = 0; k < 6; ++k) tobs[k] = obs[k];
        if (has) {
#pragma unroll
            for (int k = 0; k < OBS; ++k) tobs[k] = a.term_obs[((size_t)pend_t * N + n) * OBS + k];
        }
        float ev[4] = {0.f, 0.f, 0.f, 0.f};
        net.template forward<false, true, !JOINT>(tobs, ev);
        if (has) a.ev_dense[(size_t)pend_t * N + n] = ev[3] + net.bias4[3];
        pend_t = -1;
    };
    SampleDraws draws;
    draws.w0 = draws.w1 = draws.w2 = 0u;
    uint32_t flag_bits = 0u;
    auto step_body = [&](auto general, const int t) __attribute__((always_inline)) {
        constexpr bool GENERAL = decltype(general)::value;     // (only ever the rollout's first step: nothing is pending)
        StepOut o;
        o.trunc = 0;
        // the step's uniform draw first (see sample_uniform)
        const float u_draw = draws.next(a.env.seed, a.env.env_base + nn, a.sample_step0 + (uint64_t)t, t == 0);
        // ---- backbone(s) + heads (src/ppo.py:126-145, :174-182)
        float out[4] = {0.f, 0.f, 0.f, 0.f};
        if (!(ABLATE(a) & 1)) {
            if (JOINT) {
                net.template forward<true, true, false>(obs, out);
            } else {
                net.template forward<true, false, false>(obs, out);
                net.template forward<false, true, true>(obs, out);
            }
        }
        float lg[A];
        if constexpr (Net::DIFF) {                 // two actions, logits known up to a shift: (0, l1 - l0)
            lg[0] = 0.0f;
            lg[1] = out[0] + net.bias4[0];
        } else {
#pragma unroll
            for (int j = 0; j < A; ++j) lg[j] = out[j] + net.bias4[j];
        }
        const float value = out[3] + net.bias4[3];

        int action;
        float logp;
        if (!(ABLATE(a) & 4)) {
            categorical_from_uniform(lg, A, u_draw, action, logp);
        } else {
            action = lg[0] > lg[1]; logp = lg[0];
        }

        if constexpr (PAIR) {                      // both lanes of a pair step their copy of the env, with ITS action
            action = __shfl(action, lane & 31, 64);
            if (!(ABLATE(a) & 2))
                env_transition<Env, DRAWS, true, GENERAL>(e, aux, action, a.env, nn, t == a.T - 1, o, DRAWS ? nullptr : &ahead, live);
        }
        if (live) {
            if (PAIR && !(ABLATE(a) & 2)) {
                // (stepped above)
            } else if (!(ABLATE(a) & 2)) {
                env_transition<Env, DRAWS, false, GENERAL>(e, aux, action, a.env, nn, t == a.T - 1, o, DRAWS ? nullptr : &ahead);
            } else {
#pragma unroll
                for (int k = 0; k < 6; ++k) o.obs[k] = obs[k] + 1e-3f * (float)action;
                o.reward = 1.0f; o.done = 0; o.trunc = (t == a.T - 1);
#pragma unroll
                for (int k = 0; k < 6; ++k) o.term[k] = o.obs[k];
            }
            const size_t row = (size_t)t * N + n;
            // states / actions / log-probs are next read by the update phase: stream them past the
            // caches; rewards, values and flags are what the GAE kernel reads right after this kernel
            store_obs_stream<Env>(a.states, row, obs);  // pre-step observation (src/ppo.py:390)
            __builtin_nontemporal_store(action, a.actions + row);
            __builtin_nontemporal_store(logp, a.log_probs + row);
            a.values[row] = value;
            a.rewards[row] = o.reward;
            rsum += o.reward;
            // the i32 flag arrays are the API's format; the GAE that follows a rollout in the trainer reads the flag
            // words below, so these stream past the caches too (A/B at C2, three rounds: 477.9 -> 476.0 us per step)
            __builtin_nontemporal_store(o.done, a.done + row);
            __builtin_nontemporal_store(o.trunc, a.trunc + row);
            flag_bits |= (uint32_t)(o.done | (o.trunc << 1)) << (2 * (t & 15));
            if ((t & 15) == 15 || t == a.T - 1) {   // sixteen steps of flags per word
                if (a.flag_words) a.flag_words[(size_t)(t >> 4) * N + n] = flag_bits;
                flag_bits = 0u;
            }
            if (o.done + o.trunc > 0) store_obs<Env>(a.term_obs, row, o.term);
        }
        const bool need_ev = a.ev_dense != nullptr && live && o.trunc > 0;
        // ---- bootstrap value of the terminal observation for truncated episodes: what the
        // reference gets from agent.get_values(end_states) (src/ppo.py:555) and uses where
        // trunc != 0 (:478).  A second network pass for the whole wave -- so it is deferred: an env
        // remembers the row of its truncation (the terminal observation is in term_obs already), and
        // the pass runs when an env truncates again before its earlier one is settled (at the forced
        // truncation of the last row, for envs that also hit the time limit earlier) and once after
        // the loop: one or two passes per rollout instead of one on every step on which some env
        // of the wave reaches the 500-step limit (~1/8 of the steps for long episodes).
        if constexpr (!GENERAL) {
            if (__any(pend_t >= 0 && need_ev)) settle();
        }
        if (need_ev) pend_t = t;
        if (live) {
#pragma unroll
            for (int k = 0; k < 6; ++k) obs[k] = o.obs[k];
        }
    };
    int t0 = 0;
    if constexpr (Env::TWO_FORMS) {
        if (general_form) {                        // (T >= 1: ppo_rollout_fused returns before the launch otherwise)
            step_body(std::true_type{}, 0);
            t0 = 1;
        }
    }
    for (int t = t0; t < a.T; ++t) step_body(std::false_type{}, t);
    TL_MARK(3);
    if (__any(pend_t >= 0)) settle();
    TL_MARK(4);
    if (live) {
        store_env(e, a.env, n);
        store_obs<Env>(a.obs_io, n, obs);
    }
    if (a.reward_sum) {                            // total_return of the rollout (src/ppo.py:318, :348)
        const double w = wave_sum(live ? (double)rsum : 0.0);
        if (lane == 0) atomicAdd(a.reward_sum, w);
    }
    TL_MARK(5);
}

template <class Env, class Net> void launch_net(const RolloutArgs &a, unsigned grid, hipStream_t s)
{
    if (a.env.reset_draw) {
        PPO_LAUNCH((k_rollout<Env, Net, true>), dim3(grid), dim3(ROLLOUT_BLOCK), s, a, a.params);
    } else if constexpr (Net::FULL_OK) {
        // (every wave of the GRID must be full: N a multiple of the envs per workgroup, not just per wave)
        if (a.env.N % ROLLOUT_BLOCK == 0) PPO_LAUNCH((k_rollout<Env, Net, false, true>), dim3(grid), dim3(ROLLOUT_BLOCK), s, a, a.params);
        else PPO_LAUNCH((k_rollout<Env, Net, false>), dim3(grid), dim3(ROLLOUT_BLOCK), s, a, a.params);
    } else {
        PPO_LAUNCH((k_rollout<Env, Net, false>), dim3(grid), dim3(ROLLOUT_BLOCK), s, a, a.params);
    }
}

// Envs per wave for the f16-split nets: 64, or 32 when the batch is too small to give every SIMD of
// the chip (256 CUs x 4) a wave of 64 -- twice the waves, each with half the network work.
// wave_envs: 0 = choose by N, 32 / 64 = forced (tests and tuning, through ppo_rollout_fused_ex).
constexpr int HALF_WAVE_MAX_ENVS = 32768;

template <class Env, int A> int launch_rollout(const RolloutArgs &a, int joint, int net_mode, int wave_envs, hipStream_t s)
{
    constexpr int OBS = Env::OBS;
    const auto grid_for = [&](int wave_envs) {
        const int per_block = wave_envs * (ROLLOUT_BLOCK / 64);
        return (unsigned)((a.env.N + per_block - 1) / per_block);
    };
    const bool half_waves = wave_envs ? wave_envs == 32 : a.env.N <= HALF_WAVE_MAX_ENVS;
    if (net_mode == PPO_NET_F32) {
        if (joint) launch_net<Env, NetF32<OBS, A, true>>(a, grid_for(64), s);
        else launch_net<Env, NetF32<OBS, A, false>>(a, grid_for(64), s);
    } else if (net_mode == PPO_NET_BF16X3) {
        if (half_waves) {
            if (joint) launch_net<Env, NetF16<SchemeBF16, OBS, A, true, 1>>(a, grid_for(32), s);
            else launch_net<Env, NetF16<SchemeBF16, OBS, A, false, 1>>(a, grid_for(32), s);
        } else {
            if (joint) launch_net<Env, NetF16<SchemeBF16, OBS, A, true, 2>>(a, grid_for(64), s);
            else launch_net<Env, NetF16<SchemeBF16, OBS, A, false, 2>>(a, grid_for(64), s);
        }
    } else {
        if (half_waves) {
            if (joint) launch_net<Env, NetF16<SchemeF16, OBS, A, true, 1>>(a, grid_for(32), s);
            else launch_net<Env, NetF16<SchemeF16, OBS, A, false, 1>>(a, grid_for(32), s);
        } else {
            if (joint) launch_net<Env, NetF16<SchemeF16, OBS, A, true, 2>>(a, grid_for(64), s);
            else launch_net<Env, NetF16<SchemeF16, OBS, A, false, 2>>(a, grid_for(64), s);
        }
    }
    return ppo_launch_status();
}

}  // namespace

extern "C" {

size_t ppo_rollout_param_floats(int kind, int joint, int net_mode)
{
    const int obs = ppo_env_obs_dim(kind);
    if (obs < 0 || net_mode < PPO_NET_F32 || net_mode > PPO_NET_BF16X3) return 0;
    const size_t bb = net_mode == PPO_NET_F32 ? (size_t)backbone_floats(obs) : (size_t)bb16_words(net_mode == PPO_NET_F16_SPLIT ? 2 : 3);
    return bb * (joint ? 1 : 2) + (size_t)HEAD_FLOATS;
}

// Private form (tests and tuning; not in the public header): ppo_rollout_fused with the number of
// envs per wave of the f16-split kernels forced to 32 or 64 (0 = chosen by N, what the public entry does).
int ppo_rollout_fused_ex(int kind, double *state, int *elapsed, float *ep_ret, int *ep_len, uint32_t *episode,
                         const double *reset_draw, uint64_t seed, uint64_t env_base, const float *params,
                         int joint, int net_mode, uint64_t sample_step0, float *obs_io, float *states, int *actions,
                         float *log_probs, float *values, float *rewards, int *done, int *trunc, float *term_obs,
                         float *ev_dense, uint32_t *flag_words, double *reward_sum, float *ret_ring, int *len_ring, int *ring_cnt,
                         int ring, int N, int T, int wave_envs, void *stream)
{
    PPO_REQUIRE(wave_envs == 0 || wave_envs == 32 || wave_envs == 64);
    PPO_REQUIRE(kind == PPO_ENV_CARTPOLE || kind == PPO_ENV_ACROBOT);
    PPO_REQUIRE(!reward_sum || ppo_aligned(reward_sum, 8));
    PPO_REQUIRE(net_mode >= PPO_NET_F32 && net_mode <= PPO_NET_BF16X3);
    PPO_REQUIRE(N >= 0 && T >= 0);
    if (N == 0 || T == 0) return 0;
    PPO_REQUIRE(state && elapsed && ep_ret && ep_len && episode && params && obs_io && states && actions &&
                log_probs && values && rewards && done && trunc && term_obs);
    const size_t oa = kind == PPO_ENV_CARTPOLE ? 16 : 8;
    PPO_REQUIRE(ppo_aligned(state, 16) && ppo_aligned(obs_io, oa) && ppo_aligned(states, oa) &&
                ppo_aligned(term_obs, oa) && ppo_aligned(params, 16));
    PPO_REQUIRE(!reset_draw || ppo_aligned(reset_draw, 16));
    const bool rings = ret_ring || len_ring || ring_cnt;
    PPO_REQUIRE(!rings || (ret_ring && len_ring && ring_cnt && ring > 0));
    RolloutArgs a{};
    a.env.state = state; a.env.elapsed = elapsed; a.env.ep_ret = ep_ret; a.env.ep_len = ep_len;
    a.env.episode = episode; a.env.reset_draw = reset_draw; a.env.seed = seed; a.env.env_base = env_base;
    a.env.ret_ring = ret_ring; a.env.len_ring = len_ring; a.env.ring_cnt = ring_cnt; a.env.ring = ring;
    a.env.N = N;
    a.params = params; a.sample_step0 = sample_step0; a.obs_io = obs_io; a.states = states; a.actions = actions;
    a.log_probs = log_probs; a.values = values; a.rewards = rewards; a.done = done; a.trunc = trunc;
    a.term_obs = term_obs; a.ev_dense = ev_dense; a.flag_words = flag_words; a.reward_sum = reward_sum; a.T = T;
#ifdef PPO_ROLLOUT_ABLATE_BUILD
    static const int ablate_env = [] { const char *e = getenv("PPO_ROLLOUT_ABLATE"); return e ? atoi(e) : 0; }();
    a.ablate = ablate_env;                          // diagnostic builds only; read once
#endif
    hipStream_t s = (hipStream_t)stream;
    if (kind == PPO_ENV_CARTPOLE) return launch_rollout<CartPole, 2>(a, joint, net_mode, wave_envs, s);
    return launch_rollout<Acrobot, 3>(a, joint, net_mode, wave_envs, s);
}

int ppo_rollout_fused(int kind, double *state, int *elapsed, float *ep_ret, int *ep_len, uint32_t *episode,
                      const double *reset_draw, uint64_t seed, uint64_t env_base, const float *params,
                      int joint, int net_mode, uint64_t sample_step0, float *obs_io, float *states, int *actions,
                      float *log_probs, float *values, float *rewards, int *done, int *trunc, float *term_obs,
                      float *ev_dense, double *reward_sum, float *ret_ring, int *len_ring, int *ring_cnt, int ring,
                      int N, int T, void *stream)
{
    return ppo_rollout_fused_ex(kind, state, elapsed, ep_ret, ep_len, episode, reset_draw, seed, env_base, params, joint,
                                net_mode, sample_step0, obs_io, states, actions, log_probs, values, rewards, done, trunc,
                                term_obs, ev_dense, nullptr, reward_sum, ret_ring, len_ring, ring_cnt, ring, N, T, 0, stream);
}

// ppo_rollout_fused that also leaves the packed flag words for ppo_gae_dense_packed
int ppo_rollout_fused_packed(int kind, double *state, int *elapsed, float *ep_ret, int *ep_len, uint32_t *episode,
                             const double *reset_draw, uint64_t seed, uint64_t env_base, const float *params,
                             int joint, int net_mode, uint64_t sample_step0, float *obs_io, float *states, int *actions,
                             float *log_probs, float *values, float *rewards, int *done, int *trunc, float *term_obs,
                             float *ev_dense, uint32_t *flag_words, double *reward_sum, float *ret_ring, int *len_ring,
                             int *ring_cnt, int ring, int N, int T, void *stream)
{
    PPO_REQUIRE(flag_words || N == 0 || T == 0);
    PPO_REQUIRE(ppo_aligned(flag_words, 4));
    return ppo_rollout_fused_ex(kind, state, elapsed, ep_ret, ep_len, episode, reset_draw, seed, env_base, params, joint,
                                net_mode, sample_step0, obs_io, states, actions, log_probs, values, rewards, done, trunc,
                                term_obs, ev_dense, flag_words, reward_sum, ret_ring, len_ring, ring_cnt, ring, N, T, 0, stream);
}

}  // extern "C"

#ifdef PPO_ROLLOUT_TIMELINE
extern "C" int ppo_rollout_timeline_read(unsigned long long *dst, int words)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rollout_timeline), (size_t)words * 8, 0, hipMemcpyDeviceToHost);
}
#endif
