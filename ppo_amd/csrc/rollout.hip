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
// Per wave (64 envs) and step the two 64-wide tanh layers run on the matrix cores with
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate -- exactly a k-ordered fmaf chain, so no precision
// is given up).  The layers are CHAINED IN REGISTERS: the D layout of one 32x32 tile (lane = env
// l & 31, register r = unit (r&3) + 8 (r>>2) + 4 (l>>5)) is, register by register, a valid B
// operand of the next layer (lane = (k half l >> 5, env l & 31)) for the k-pair
// (unit(r), unit(r) + 4) -- so the next layer's weights are packed on the host in THAT k order and
// an accumulator register goes straight from tanh into the next MFMA.  No LDS round trip, no
// shuffles.  The whole network (72 A-operand fragments + the per-lane head weights) stays in the
// 512-entry register file for all T steps (one wave per SIMD), so the time-step loop of the joint
// net touches no LDS at all.  Earlier formulations, measured at N = 65536, T = 128, joint net:
//   - weights as SGPR operands via the scalar cache (VALU dot products)          5.4 ms / rollout
//   - weights broadcast from LDS with ds_read_b128 (VALU)                         2.9 ms
//   - MFMA with weights and a [unit][env] activation tile in LDS                  1.56 ms
// The heads are per-lane partial dot products over the 32 units a lane holds, completed with one
// v_permlane32_swap + add, which also hands every lane the result of ITS env (lane = env again for
// sampling (Philox) and the fp64 env physics of env_core.h).
//
// Bound: MFMA issue (144 x 64 cycles per step, joint net) + the VALU work that does not fit in
// its shadow.  HBM: 40 B written per env-step (states 16, action, log-prob, value, reward, done,
// trunc).
#include "env_core.h"
#include <cstdlib>

using namespace ppo_env;

namespace {

constexpr int H = 64;  // hidden width of the reference's dense net (src/ppo.py:127-129)

struct RolloutArgs {
    EnvArgs env;             // state, counters, rings, reset stream (action/obs_out/... unused)
    const float *params;     // packed network, layout below
    uint64_t sample_step0;   // Philox position of time step 0 of this rollout
    float *obs_io;           // [N,OBS] current observations in; carried observations out
    float *states;           // [T,N,OBS]
    int *actions;            // [T,N]
    float *log_probs, *values, *rewards;  // [T,N]
    int *done, *trunc;       // [T,N]
    float *term_obs;         // [T,N,OBS], written where a flag is raised
    float *ev_dense;         // [T,N] or NULL: V(terminal obs) written where trunc != 0
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
__host__ __device__ constexpr int ksteps(int k_in) { return 1 + (k_in + 1) / 2; }
__host__ __device__ constexpr int backbone_floats(int obs) { return (ksteps(obs) + ksteps(H)) * 2 * 64; }
constexpr int HEAD_FLOATS = 4 * 2 * 16 * 64 + 4;

using f32x16 = __attribute__((ext_vector_type(16))) float;

// tanh(x) = sign(x) * (1 - 2 / (exp(2|x|) + 1)), branch-free on the hardware exp2 / rcp units
// (6 VALU instructions; ocml's tanhf is two divergent branches of ~20).  Absolute error < 2e-7,
// i.e. the size of one f32 rounding of an O(1) activation; exp2 -> inf gives exactly +-1.
__device__ __forceinline__ float fast_tanh(float x)
{
    const float t = __builtin_amdgcn_exp2f(2.8853900817779268f * __builtin_fabsf(x));  // exp(2|x|)
    const float r = 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
    return __builtin_copysignf(r, x);
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
    // ---- layer 2, one output tile at a time: tanh(h1) registers are the B operands as they are.
    // The heads of output tile 0 have no dependence on tile 1's MFMAs and fill their issue gaps.
    float p[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j][0] = p[j][1] = 0.0f;
    f32x16 h2[2][2];
#pragma unroll
    for (int op = 0; op < 2; ++op) {
        h2[op][0] = PPO_MFMA(w2.get(0, op), one, zero16());
        h2[op][1] = PPO_MFMA(w2.get(0, op), one, zero16());
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (op == 0) {
                    h1[o][0][r] = fast_tanh(h1[o][0][r]);
                    h1[o][1][r] = fast_tanh(h1[o][1][r]);
                }
                const float w = w2.get(1 + 16 * o + r, op);
                h2[op][0] = PPO_MFMA(w, h1[o][0][r], h2[op][0]);
                h2[op][1] = PPO_MFMA(w, h1[o][1][r], h2[op][1]);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float h = fast_tanh(h2[op][e][r]);
                if (DO_P) {
#pragma unroll
                    for (int j = 0; j < A; ++j) p[j][e] = __builtin_fmaf(h, hl[j][op][r], p[j][e]);
                }
                if (DO_V) p[3][e] = __builtin_fmaf(h, hl[3][op][r], p[3][e]);
            }
    }
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

template <class Env, int A, bool JOINT>
__global__ void __launch_bounds__(ROLLOUT_BLOCK) k_rollout(RolloutArgs a, const float *__restrict__ params)
{
    constexpr int OBS = Env::OBS;
    static_assert(OBS % 2 == 0 && A <= 3, "fragment layout assumes an even input width and <= 3 actions");
    constexpr int NB = backbone_floats(OBS);
    constexpr int S1 = ksteps(OBS), S2 = ksteps(H);

    const int lane = threadIdx.x & 63, half = lane >> 5;
    const size_t n = (size_t)blockIdx.x * ROLLOUT_BLOCK + threadIdx.x;
    // Tail lanes (n >= N) stay: MFMA needs all 64 lanes.  They shadow env N-1 and never store.
    const bool live = n < (size_t)a.env.N;
    const size_t nn = live ? n : (size_t)a.env.N - 1;

    // ---- the network: policy (or joint) backbone in registers for the whole rollout; the value
    // backbone of split nets is shared by the block's waves through LDS.
    RegFrags<S1> P1;
    RegFrags<S2> P2;
    P1.load(params, lane);
    P2.load(params + S1 * 128, lane);
    __shared__ float sV[JOINT ? 1 : NB];
    if (!JOINT) {
        for (int k = threadIdx.x; k < NB; k += ROLLOUT_BLOCK) sV[k] = params[NB + k];
        __syncthreads();                           // the only barrier: waves are independent from here on
    }
    const LdsFrags V1{sV + lane}, V2{sV + (JOINT ? 0 : S1 * 128) + lane};
    const float *heads = params + NB * (JOINT ? 1 : 2);
    float hl[4][2][16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) hl[j][o][r] = (j < A || j == 3) ? heads[((j * 2 + o) * 16 + r) * 64 + lane] : 0.0f;
    float bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bias4[j] = heads[4 * 2 * 16 * 64 + j];

    EnvRegs e;
    load_env(e, a.env, nn);
    float obs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < OBS; ++k) obs[k] = a.obs_io[nn * OBS + k];

    const size_t N = (size_t)a.env.N;
    for (int t = 0; t < a.T; ++t) {
        // ---- backbone(s) + heads (src/ppo.py:126-145, :174-182)
        float out[4] = {0.f, 0.f, 0.f, 0.f};
        if (!(a.ablate & 1)) {
            if (JOINT) {
                net_forward<OBS, A, true, true>(obs, P1, P2, hl, half, out);
            } else {
                net_forward<OBS, A, true, false>(obs, P1, P2, hl, half, out);
                net_forward<OBS, A, false, true>(obs, V1, V2, hl, half, out);
            }
        }
        float lg[A];
#pragma unroll
        for (int j = 0; j < A; ++j) lg[j] = out[j] + bias4[j];
        const float value = out[3] + bias4[3];

        int action;
        float logp;
        if (!(a.ablate & 4)) {
            categorical_sample(lg, A, a.env.seed, a.env.env_base + nn, a.sample_step0 + (uint64_t)t, action, logp);
        } else {
            action = lg[0] > lg[1]; logp = lg[0];
        }

        StepOut o;
        o.trunc = 0;
        if (live) {
            if (!(a.ablate & 2)) {
                env_transition<Env>(e, action, a.env, nn, t == a.T - 1, o);
            } else {
#pragma unroll
                for (int k = 0; k < 6; ++k) o.obs[k] = obs[k] + 1e-3f * (float)action;
                o.reward = 1.0f; o.done = 0; o.trunc = (t == a.T - 1);
#pragma unroll
                for (int k = 0; k < 6; ++k) o.term[k] = o.obs[k];
            }
            const size_t row = (size_t)t * N + n;
            store_obs<Env>(a.states, row, obs);        // pre-step observation (src/ppo.py:390)
            a.actions[row] = action;
            a.log_probs[row] = logp;
            a.values[row] = value;
            a.rewards[row] = o.reward;
            a.done[row] = o.done;
            a.trunc[row] = o.trunc;
            if (o.done + o.trunc > 0) store_obs<Env>(a.term_obs, row, o.term);
        }
        // ---- bootstrap value of the terminal observation for truncated episodes: what the
        // reference gets from agent.get_values(end_states) (src/ppo.py:555) and uses where
        // trunc != 0 (:478).  Wave-uniform branch: taken when any of the 64 envs truncated (always
        // at the forced last step, ~1/8 of the other steps for a policy that reaches the 500-step limit).
        const bool need_ev = a.ev_dense != nullptr && live && o.trunc > 0;
        if (__any(need_ev)) {
            float tobs[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) tobs[k] = need_ev ? o.term[k] : obs[k];
            float ev[4] = {0.f, 0.f, 0.f, 0.f};
            if (JOINT) net_forward<OBS, A, false, true>(tobs, P1, P2, hl, half, ev);
            else net_forward<OBS, A, false, true>(tobs, V1, V2, hl, half, ev);
            if (need_ev) a.ev_dense[(size_t)t * N + n] = ev[3] + bias4[3];
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < 6; ++k) obs[k] = o.obs[k];
        }
    }
    if (live) {
        store_env(e, a.env, n);
        store_obs<Env>(a.obs_io, n, obs);
    }
}

template <class Env, int A> int launch_rollout(const RolloutArgs &a, int joint, hipStream_t s)
{
    const unsigned grid = (unsigned)((a.env.N + ROLLOUT_BLOCK - 1) / ROLLOUT_BLOCK);
    if (joint) hipLaunchKernelGGL((k_rollout<Env, A, true>), dim3(grid), dim3(ROLLOUT_BLOCK), 0, s, a, a.params);
    else hipLaunchKernelGGL((k_rollout<Env, A, false>), dim3(grid), dim3(ROLLOUT_BLOCK), 0, s, a, a.params);
    return ppo_launch_status();
}

}  // namespace

extern "C" {

size_t ppo_rollout_param_floats(int kind, int joint)
{
    const int obs = ppo_env_obs_dim(kind);
    if (obs < 0) return 0;
    const int A = kind == PPO_ENV_CARTPOLE ? 2 : 3;
    (void)A;
    return (size_t)backbone_floats(obs) * (joint ? 1 : 2) + (size_t)HEAD_FLOATS;
}

int ppo_rollout_fused(int kind, double *state, int *elapsed, float *ep_ret, int *ep_len, uint32_t *episode,
                      const double *reset_draw, uint64_t seed, uint64_t env_base, const float *params,
                      int joint, uint64_t sample_step0, float *obs_io, float *states, int *actions,
                      float *log_probs, float *values, float *rewards, int *done, int *trunc, float *term_obs,
                      float *ev_dense, float *ret_ring, int *len_ring, int *ring_cnt, int ring, int N, int T, void *stream)
{
    PPO_REQUIRE(kind == PPO_ENV_CARTPOLE || kind == PPO_ENV_ACROBOT);
    PPO_REQUIRE(N >= 0 && T >= 0);
    if (N == 0 || T == 0) return 0;
    PPO_REQUIRE(state && elapsed && ep_ret && ep_len && episode && params && obs_io && states && actions &&
                log_probs && values && rewards && done && trunc && term_obs);
    const size_t oa = kind == PPO_ENV_CARTPOLE ? 16 : 8;
    PPO_REQUIRE(ppo_aligned(state, 16) && ppo_aligned(obs_io, oa) && ppo_aligned(states, oa) &&
                ppo_aligned(term_obs, oa) && ppo_aligned(params, 4));
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
    a.term_obs = term_obs; a.ev_dense = ev_dense; a.T = T;
    const char *abl = getenv("PPO_ROLLOUT_ABLATE");
    a.ablate = abl ? atoi(abl) : 0;
    hipStream_t s = (hipStream_t)stream;
    if (kind == PPO_ENV_CARTPOLE) return launch_rollout<CartPole, 2>(a, joint, s);
    return launch_rollout<Acrobot, 3>(a, joint, s);
}

}  // extern "C"
