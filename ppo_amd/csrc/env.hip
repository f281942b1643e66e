// Batched CartPole-v1 / Acrobot-v1: fused step + TimeLimit + episode statistics +
// forced last-step truncation + terminal-observation capture + auto-reset + obs write.
// One lane per env, everything in registers -- gfx950.
//
// Replaces SyncVecEnv.vec_step / env_reset / vec_reset (reference src/ppo.py:269-302),
// the per-actor loop of SyncVecEnv.rollout (:342-388) and the env physics the reference
// gets from gymnasium 0.29.1 (call sites src/env_cartpole.py:7-8, src/env_acrobot.py:7-8;
// algorithm restated in SURVEY.md 8c).  State is fp64 like gymnasium's, observations are
// the f32 cast; every operation rounds once (-ffp-contract=off) in the source order of
// the Python expressions, so + - * / agree bit for bit with CPython/numpy and only
// sin/cos (device libm vs glibc) can differ, by an ulp.
//
// HBM traffic: 108 B (CartPole) / 116 B (Acrobot) per env-step; at N = 65536 that is
// ~7 MB per launch, i.e. the kernel is launch/latency-bound, not bandwidth-bound.
// Philox4x32-10 keyed (seed; global env id, reset counter) feeds resets, so results do
// not depend on how envs are sharded over GPUs.
#include "env_core.h"

using namespace ppo_env;

namespace {

template <class Env> __global__ void __launch_bounds__(256) k_env_reset(EnvArgs a)
{
    const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (size_t)a.N) return;
    const uint32_t ep = a.episode[n];
    double u[4];
    EnvRegs e;
    float obs[6];
    draw_reset(a, n, ep, u);
    typename Env::Aux aux;
    Env::reset(e.s, u, obs, aux);
    e.elapsed = 0; e.ep_ret = 0.0f; e.ep_len = 0;
    store_env(e, a, n);
    a.episode[n] = ep + 1;
    store_obs<Env>(a.obs_out, n, obs);
}

template <class Env> __global__ void __launch_bounds__(256) k_env_step(EnvArgs a)
{
    const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (size_t)a.N) return;
    EnvRegs e;
    load_env(e, a, n);
    StepOut o;
    typename Env::Aux aux;
    if (Env::TWO_FORMS && Env::wide(e.s)) {        // a caller-written state outside what a transition leaves (env_core.h)
        Env::template prime<false, true>(e.s, aux);
        env_transition<Env, true, false, true>(e, aux, a.action[n], a, n, a.is_last, o);
    } else {
        Env::prime(e.s, aux);
        env_transition<Env>(e, aux, a.action[n], a, n, a.is_last, o);
    }
    a.reward[n] = o.reward;
    a.done[n] = o.done;
    a.trunc[n] = o.trunc;
    if (a.term_obs && o.done + o.trunc > 0) store_obs<Env>(a.term_obs, n, o.term);
    store_env(e, a, n);
    store_obs<Env>(a.obs_out, n, o.obs);
}

constexpr int MAX_ACTIONS = 32;
__global__ void __launch_bounds__(256) k_categorical_sample(const float *logits, int A, uint64_t seed,
                                                            uint64_t env_base, uint64_t step, int *action,
                                                            float *logp, int N)
{
    const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (size_t)N) return;
    int act;
    float lp;
    categorical_sample(logits + n * (size_t)A, A, seed, env_base + n, step, act, lp);
    action[n] = act;
    logp[n] = lp;
}

// ---- test hooks (private; tests/test_gpu_env.py): pieces of the Acrobot transition on caller-made operands ----
// flag[i] = Acrobot::terminal of state[i] (what the step kernels decide after a transition); lhs[i] = the value the
// kernel's fast path compares with 1.0 (cos(theta1 + theta2) by the angle-sum identity)
__global__ void __launch_bounds__(256) k_test_acrobot_terminal(const double *state, int *flag, double *lhs, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long j = i < n ? i : n - 1;                   // every lane runs (the guard band's branch is wave-uniform)
    const double s[4] = {state[4 * j], state[4 * j + 1], state[4 * j + 2], state[4 * j + 3]};
    const Acrobot::Trig t = Acrobot::trig_of(s[0], s[1]);
    const bool term = Acrobot::terminal(s, t);
    if (i < n) {
        flag[i] = term ? 1 : 0;
        lhs[i] = -t.c1 - __builtin_fma(t.c1, t.c2, -(t.s1 * t.s2));
    }
}

// q_fast[i] = Acrobot::quot(num[i], den[i], Acrobot::recip(den[i])) -- the division the RK4 stages use --
// and q_ieee[i] = num[i] / den[i] as the compiler's IEEE division sequence computes it
__global__ void __launch_bounds__(256) k_test_acrobot_quot(const double *num, const double *den, double *q_fast,
                                                           double *q_ieee, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = num[i], d = den[i];
    q_fast[i] = Acrobot::quot(a, d, Acrobot::recip(d));
    q_ieee[i] = a / d;
}

int env_dispatch(int kind, bool is_reset, const EnvArgs &a, hipStream_t s)
{
    const int block = 256;
    const unsigned grid = (unsigned)((a.N + block - 1) / block);
    if (kind == PPO_ENV_CARTPOLE) {
        if (is_reset) hipLaunchKernelGGL(k_env_reset<CartPole>, dim3(grid), dim3(block), 0, s, a);
        else hipLaunchKernelGGL(k_env_step<CartPole>, dim3(grid), dim3(block), 0, s, a);
    } else {
        if (is_reset) hipLaunchKernelGGL(k_env_reset<Acrobot>, dim3(grid), dim3(block), 0, s, a);
        else hipLaunchKernelGGL(k_env_step<Acrobot>, dim3(grid), dim3(block), 0, s, a);
    }
    return ppo_launch_status();
}

}  // namespace

extern "C" {

int ppo_abi_version(void) { return PPO_AMD_ABI_VERSION; }

int ppo_env_obs_dim(int kind) { return kind == PPO_ENV_CARTPOLE ? 4 : (kind == PPO_ENV_ACROBOT ? 6 : -1); }

int ppo_env_reset(int kind, double *state, int *elapsed, float *ep_ret, int *ep_len, uint32_t *episode,
                  const double *reset_draw, uint64_t seed, uint64_t env_base, float *obs_out, int N,
                  void *stream)
{
    PPO_REQUIRE(kind == PPO_ENV_CARTPOLE || kind == PPO_ENV_ACROBOT);
    PPO_REQUIRE(N >= 0);
    if (N == 0) return 0;
    PPO_REQUIRE(state && elapsed && ep_ret && ep_len && episode && obs_out);
    PPO_REQUIRE(ppo_aligned(state, 16) && ppo_aligned(obs_out, kind == PPO_ENV_CARTPOLE ? 16 : 8));
    PPO_REQUIRE(!reset_draw || ppo_aligned(reset_draw, 16));
    EnvArgs a{};
    a.state = state; a.elapsed = elapsed; a.ep_ret = ep_ret; a.ep_len = ep_len; a.episode = episode;
    a.reset_draw = reset_draw; a.seed = seed; a.env_base = env_base; a.obs_out = obs_out; a.N = N;
    return env_dispatch(kind, true, a, (hipStream_t)stream);
}

int ppo_env_step(int kind, double *state, int *elapsed, float *ep_ret, int *ep_len, uint32_t *episode,
                 const int *action, const double *reset_draw, uint64_t seed, uint64_t env_base,
                 float *obs_out, float *term_obs, float *reward, int *done, int *trunc, float *ret_ring,
                 int *len_ring, int *ring_cnt, int ring, int N, int is_last, void *stream)
{
    PPO_REQUIRE(kind == PPO_ENV_CARTPOLE || kind == PPO_ENV_ACROBOT);
    PPO_REQUIRE(N >= 0);
    if (N == 0) return 0;
    PPO_REQUIRE(state && elapsed && ep_ret && ep_len && episode && action && obs_out && reward && done && trunc);
    const size_t oa = kind == PPO_ENV_CARTPOLE ? 16 : 8;
    PPO_REQUIRE(ppo_aligned(state, 16) && ppo_aligned(obs_out, oa) && (!term_obs || ppo_aligned(term_obs, oa)));
    PPO_REQUIRE(!reset_draw || ppo_aligned(reset_draw, 16));
    const bool rings = ret_ring || len_ring || ring_cnt;
    PPO_REQUIRE(!rings || (ret_ring && len_ring && ring_cnt && ring > 0));
    EnvArgs a{};
    a.state = state; a.elapsed = elapsed; a.ep_ret = ep_ret; a.ep_len = ep_len; a.episode = episode;
    a.action = action; a.reset_draw = reset_draw; a.seed = seed; a.env_base = env_base;
    a.obs_out = obs_out; a.term_obs = term_obs; a.reward = reward; a.done = done; a.trunc = trunc;
    a.ret_ring = ret_ring; a.len_ring = len_ring; a.ring_cnt = ring_cnt; a.ring = ring; a.N = N;
    a.is_last = is_last;
    return env_dispatch(kind, false, a, (hipStream_t)stream);
}

int ppo_test_acrobot_terminal(const double *state, int *flag, double *lhs, long n, void *stream)
{
    PPO_REQUIRE(state && flag && lhs && n > 0);
    hipLaunchKernelGGL(k_test_acrobot_terminal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       state, flag, lhs, n);
    return ppo_launch_status();
}

int ppo_test_acrobot_quot(const double *num, const double *den, double *q_fast, double *q_ieee, long n, void *stream)
{
    PPO_REQUIRE(num && den && q_fast && q_ieee && n > 0);
    hipLaunchKernelGGL(k_test_acrobot_quot, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       num, den, q_fast, q_ieee, n);
    return ppo_launch_status();
}

int ppo_categorical_sample(const float *logits, int A, uint64_t seed, uint64_t env_base, uint64_t step,
                           int *action, float *logp, int N, void *stream)
{
    PPO_REQUIRE(N >= 0 && A >= 1 && A <= MAX_ACTIONS);
    if (N == 0) return 0;
    PPO_REQUIRE(logits && action && logp);
    hipLaunchKernelGGL(k_categorical_sample, dim3((unsigned)((N + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, logits, A, seed, env_base, step, action, logp, N);
    return ppo_launch_status();
}

}  // extern "C"
