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
// Here the activations never leave the CU.
//
// Per wave (64 envs) and step: the two 64-wide tanh layers run on the matrix cores with
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate -- exactly a k-ordered fmaf chain, so no precision
// is given up); activations live in a wave-private LDS tile [unit][env], weights are staged once
// per workgroup into LDS pre-arranged as MFMA A-operand fragments, so ONE conflict-free
// ds_read_b32 hands the wave 64 distinct weights.  That operand delivery is the point of using
// MFMA here, not its FLOP rate (equal to v_pk_fma_f32): two VALU formulations were measured first
//   - weights as SGPR operands via the scalar cache: 80 live SGPRs spill to VGPR lanes, the 17 KB
//     network thrashes the 16 KB scalar cache                                  5.4 ms / rollout
//   - weights broadcast from LDS with ds_read_b128: every wave re-reads the whole matrix through
//     the CU's one LDS port, 4 cycles per 4 weights                            2.9 ms / rollout
// Heads, sampling (Philox) and the fp64 env physics of env_core.h run lane = env on the VALU.
// After the initial weight staging no barrier exists: waves never exchange data.
//
// Bound: MFMA issue (144 x 64 cycles per step, joint net) + ~2k VALU instructions per step.
// HBM: 40 B written per env-step (states 16, action, log-prob, value, reward, done, trunc).
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
//   backbone(pi): F1[S1][2][64]  F2[33][2][64]      MFMA A-operand fragments, see frag()
//   backbone(va): same (split nets only)
//   heads:        HP[64][4] (logit weights per hidden unit, zero padded)  bp[4]
//                 HV[64]   (critic weights)                                bc[1]
// frag(): a Linear(K -> 64) layer y = W x + b is evaluated as 1 + K/2 k-steps of
// v_mfma_f32_32x32x2_f32 per 32-output tile o.  Lane l of k-step s holds
//   s = 0 : (l < 32) ? b[32 o + l] : 0                 (paired with the constant B row [1; 0])
//   s > 0 : W[32 o + (l & 31)][2 (s - 1) + (l >> 5)]
// so one conflict-free ds_read_b32 delivers 64 distinct weights to the wave.
__host__ __device__ constexpr int ksteps(int k_in) { return 1 + (k_in + 1) / 2; }
__host__ __device__ constexpr int backbone_floats(int obs) { return (ksteps(obs) + ksteps(H)) * 2 * 64; }
constexpr int HEAD_FLOATS = H * 4 + 4 + H + 1;

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

// One dense layer K_IN -> 64 with tanh for the 64 envs of this wave, in place in the wave's LDS
// tile hw[unit][env]: reads rows 0..K_IN-1, writes rows 0..63.
//   A operand (weights): lane l = (out row l & 31, k half l >> 5)  -> frag rows above
//   B operand (inputs):  lane l = (k half l >> 5, env l & 31)       -> hw[(2(s-1) + half)][32 e + (l & 31)]
//   D (32 out x 32 env): lane l holds env l & 31, register r holds out row (r&3) + 8 (r>>2) + 4 (l>>5)
// The bias k-step comes first, so every output is the fmaf chain b, +w0 x0, +w1 x1, ... in input
// order (f32 MFMA is exactly a k-ordered fmaf chain on gfx950).
template <int K_IN>
__device__ __forceinline__ void dense_tanh_mfma(float *hw, const float *frag, int lane)
{
    constexpr int S = ksteps(K_IN);
    const int half = lane >> 5, l31 = lane & 31;
    f32x16 acc[2][2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[o][e][r] = 0.0f;
    {
        const float one = half == 0 ? 1.0f : 0.0f;
        const float a0 = frag[lane], a1 = frag[64 + lane];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, one, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, one, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, one, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, one, acc[1][1], 0, 0, 0);
    }
    const float *brow = hw + half * 64 + l31;
#pragma unroll 4
    for (int s = 1; s < S; ++s) {
        const float a0 = frag[(s * 2 + 0) * 64 + lane], a1 = frag[(s * 2 + 1) * 64 + lane];
        const float b0 = brow[(s - 1) * 128], b1 = brow[(s - 1) * 128 + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int out = 32 * o + (r & 3) + 8 * (r >> 2) + 4 * half;
                hw[out * 64 + 32 * e + l31] = fast_tanh(acc[o][e][r]);
            }
}

constexpr int ROLLOUT_BLOCK = 256;  // 4 waves share one LDS copy of the network

template <class Env, int A, bool JOINT>
__global__ void __launch_bounds__(ROLLOUT_BLOCK) k_rollout(RolloutArgs a, const float *__restrict__ params)
{
    constexpr int OBS = Env::OBS;
    static_assert(OBS % 2 == 0 && A <= 3, "fragment layout assumes an even input width and <= 3 actions");
    constexpr int NB = backbone_floats(OBS);
    constexpr int NP = NB * (JOINT ? 1 : 2) + HEAD_FLOATS;
    constexpr int NP_PAD = (NP + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float sW[NP_PAD];                         // the network
    __shared__ __attribute__((aligned(16))) float sH[(ROLLOUT_BLOCK / 64) * H * 64];  // [wave][unit][env]
    for (int k = threadIdx.x; k < NP; k += ROLLOUT_BLOCK) sW[k] = params[k];
    __syncthreads();                               // the only barrier: waves are independent from here on

    const int lane = threadIdx.x & 63;
    const size_t n = (size_t)blockIdx.x * ROLLOUT_BLOCK + threadIdx.x;
    // Tail lanes (n >= N) stay: MFMA needs all 64 lanes.  They shadow env N-1 and never store.
    const bool live = n < (size_t)a.env.N;
    const size_t nn = live ? n : (size_t)a.env.N - 1;
    float *hw = sH + (threadIdx.x >> 6) * (H * 64);   // this wave's [unit][env] tile
    float *hl = hw + lane;                            // this lane's env column

    const float *F1 = sW, *F2 = F1 + ksteps(OBS) * 128;
    const float *G1 = sW + NB, *G2 = G1 + ksteps(OBS) * 128;      // value backbone (split nets)
    const float *HP = sW + NB * (JOINT ? 1 : 2), *bp = HP + H * 4, *HV = bp + 4, *bc = HV + H;

    EnvRegs e;
    load_env(e, a.env, nn);
    float obs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < OBS; ++k) obs[k] = a.obs_io[nn * OBS + k];

    const size_t N = (size_t)a.env.N;
    for (int t = 0; t < a.T; ++t) {
        // ---- policy backbone (src/ppo.py:126-131) on the matrix cores
#pragma unroll
        for (int k = 0; k < OBS; ++k) hl[k * 64] = obs[k];
        if (!(a.ablate & 1)) {
            dense_tanh_mfma<OBS>(hw, F1, lane);
            dense_tanh_mfma<H>(hw, F2, lane);
        }

        // ---- heads (:140, :145): lane = env again, 64-term fmaf chains in unit order
        float lg[A], value = bc[0];
#pragma unroll
        for (int j = 0; j < A; ++j) lg[j] = bp[j];
#pragma unroll 8
        for (int k = 0; k < H; ++k) {
            const float h = hl[k * 64];
            const float4 w = *reinterpret_cast<const float4 *>(HP + k * 4);
            lg[0] = __builtin_fmaf(h, w.x, lg[0]);
            lg[1] = __builtin_fmaf(h, w.y, lg[1]);
            if (A > 2) lg[2] = __builtin_fmaf(h, w.z, lg[2]);
            if (JOINT) value = __builtin_fmaf(h, HV[k], value);
        }
        if (!JOINT) {                              // separate value backbone (:132-138, :179-182)
#pragma unroll
            for (int k = 0; k < OBS; ++k) hl[k * 64] = obs[k];
            dense_tanh_mfma<OBS>(hw, G1, lane);
            dense_tanh_mfma<H>(hw, G2, lane);
#pragma unroll 8
            for (int k = 0; k < H; ++k) value = __builtin_fmaf(hl[k * 64], HV[k], value);
        }

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
#pragma unroll
            for (int k = 0; k < OBS; ++k) hl[k * 64] = need_ev ? o.term[k] : obs[k];
            dense_tanh_mfma<OBS>(hw, JOINT ? F1 : G1, lane);
            dense_tanh_mfma<H>(hw, JOINT ? F2 : G2, lane);
            float ev = bc[0];
#pragma unroll 8
            for (int k = 0; k < H; ++k) ev = __builtin_fmaf(hl[k * 64], HV[k], ev);
            if (need_ev) a.ev_dense[(size_t)t * N + n] = ev;
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
