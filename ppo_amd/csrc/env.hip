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
#include "common.h"

namespace {

constexpr int MAX_EPISODE_STEPS = 500;  // TimeLimit of CartPole-v1 and Acrobot-v1
constexpr double PI = 3.14159265358979323846;

struct EnvArgs {
    double *state;
    int *elapsed;
    float *ep_ret;
    int *ep_len;
    uint32_t *episode;
    const int *action;
    const double *reset_draw;
    uint64_t seed, env_base;
    float *obs_out, *term_obs, *reward;
    int *done, *trunc;
    float *ret_ring;
    int *len_ring, *ring_cnt;
    int ring, N, is_last;
};

// ---- CartPole-v1 ---------------------------------------------------------------------
struct CartPole {
    static constexpr int OBS = 4;

    __device__ static void reset(double (&s)[4], const double (&u)[4], float (&obs)[6])
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[k] = -0.05 + (0.05 - -0.05) * u[k];
            obs[k] = (float)s[k];
        }
    }

    // returns terminated; reward is always 1.0
    __device__ static bool step(double (&s)[4], int action, float (&obs)[6], float &reward)
    {
        const double gravity = 9.8, masscart = 1.0, masspole = 0.1;
        const double total_mass = masspole + masscart;
        const double length = 0.5;
        const double polemass_length = masspole * length;
        const double force_mag = 10.0, tau = 0.02;
        const double theta_thr = 12 * 2 * PI / 360;
        const double x_thr = 2.4;

        double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
        const double force = (action == 1) ? force_mag : -force_mag;
        const double costheta = cos(theta);
        const double sintheta = sin(theta);
        const double temp = (force + polemass_length * (theta_dot * theta_dot) * sintheta) / total_mass;
        const double thetaacc = (gravity * sintheta - costheta * temp) /
                                (length * (4.0 / 3.0 - masspole * (costheta * costheta) / total_mass));
        const double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
        x = x + tau * x_dot;
        x_dot = x_dot + tau * xacc;
        theta = theta + tau * theta_dot;
        theta_dot = theta_dot + tau * thetaacc;
        s[0] = x; s[1] = x_dot; s[2] = theta; s[3] = theta_dot;
        reward = 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) obs[k] = (float)s[k];
        return (x < -x_thr) || (x > x_thr) || (theta < -theta_thr) || (theta > theta_thr);
    }
};

// ---- Acrobot-v1 ("book" dynamics, RK4 over [0, 0.2] in registers) -----------------------
struct Acrobot {
    static constexpr int OBS = 6;

    __device__ static void write_obs(const double (&s)[4], float (&obs)[6])
    {
        obs[0] = (float)cos(s[0]); obs[1] = (float)sin(s[0]);
        obs[2] = (float)cos(s[1]); obs[3] = (float)sin(s[1]);
        obs[4] = (float)s[2]; obs[5] = (float)s[3];
    }

    __device__ static void reset(double (&s)[4], const double (&u)[4], float (&obs)[6])
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = (double)(float)(-0.1 + (0.1 - -0.1) * u[k]);  // .astype(float32)
        write_obs(s, obs);
    }

    __device__ static void dsdt(const double (&y)[4], double a, double (&out)[4])
    {
        const double m1 = 1.0, m2 = 1.0, l1 = 1.0, lc1 = 0.5, lc2 = 0.5, I1 = 1.0, I2 = 1.0, g = 9.8;
        const double theta1 = y[0], theta2 = y[1], dtheta1 = y[2], dtheta2 = y[3];
        const double c2 = cos(theta2), s2 = sin(theta2);
        const double d1 = m1 * (lc1 * lc1) + m2 * ((l1 * l1) + (lc2 * lc2) + 2 * l1 * lc2 * c2) + I1 + I2;
        const double d2 = m2 * ((lc2 * lc2) + l1 * lc2 * c2) + I2;
        const double phi2 = m2 * lc2 * g * cos(theta1 + theta2 - PI / 2.0);
        const double phi1 = -m2 * l1 * lc2 * (dtheta2 * dtheta2) * s2
                            - 2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * s2
                            + (m1 * lc1 + m2 * l1) * g * cos(theta1 - PI / 2)
                            + phi2;
        const double ddtheta2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * (dtheta1 * dtheta1) * s2 - phi2) /
                                (m2 * (lc2 * lc2) + I2 - (d2 * d2) / d1);
        const double ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;
        out[0] = dtheta1; out[1] = dtheta2; out[2] = ddtheta1; out[3] = ddtheta2;
    }

    __device__ static double wrap(double x, double m, double M)
    {
        const double diff = M - m;
        while (x > M) x = x - diff;
        while (x < m) x = x + diff;
        return x;
    }

    __device__ static bool step(double (&s)[4], int action, float (&obs)[6], float &reward)
    {
        const double dt = 0.2 - 0;
        const double dt2 = dt / 2.0;
        const double torque = (double)(action - 1);
        double k1[4], k2[4], k3[4], k4[4], y[4];
        dsdt(s, torque, k1);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = s[i] + dt2 * k1[i];
        dsdt(y, torque, k2);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = s[i] + dt2 * k2[i];
        dsdt(y, torque, k3);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = s[i] + dt * k3[i];
        dsdt(y, torque, k4);
        double ns[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ns[i] = s[i] + dt / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
        ns[0] = wrap(ns[0], -PI, PI);
        ns[1] = wrap(ns[1], -PI, PI);
        ns[2] = fmin(fmax(ns[2], -4 * PI), 4 * PI);
        ns[3] = fmin(fmax(ns[3], -9 * PI), 9 * PI);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = ns[i];
        const bool terminated = (-cos(ns[0]) - cos(ns[1] + ns[0]) > 1.0);
        reward = terminated ? 0.0f : -1.0f;
        write_obs(s, obs);
        return terminated;
    }
};

template <class Env> __device__ __forceinline__ void store_obs(float *dst, size_t n, const float (&obs)[6])
{
    if (Env::OBS == 4) {
        *reinterpret_cast<float4 *>(dst + n * 4) = make_float4(obs[0], obs[1], obs[2], obs[3]);
    } else {
        float2 *p = reinterpret_cast<float2 *>(dst + n * 6);
        p[0] = make_float2(obs[0], obs[1]);
        p[1] = make_float2(obs[2], obs[3]);
        p[2] = make_float2(obs[4], obs[5]);
    }
}

__device__ __forceinline__ void draw_reset(const EnvArgs &a, size_t n, uint32_t episode, double (&u)[4])
{
    if (a.reset_draw) {
        const double2 *p = reinterpret_cast<const double2 *>(a.reset_draw + n * 4);
        const double2 lo = p[0], hi = p[1];
        u[0] = lo.x; u[1] = lo.y; u[2] = hi.x; u[3] = hi.y;
    } else {
        reset_uniform4(a.seed, a.env_base + n, episode, u);
    }
}

template <class Env> __global__ void __launch_bounds__(256) k_env_reset(EnvArgs a)
{
    const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (size_t)a.N) return;
    const uint32_t ep = a.episode[n];
    double u[4], s[4];
    float obs[6];
    draw_reset(a, n, ep, u);
    Env::reset(s, u, obs);
    double2 *sp = reinterpret_cast<double2 *>(a.state + n * 4);
    sp[0] = make_double2(s[0], s[1]);
    sp[1] = make_double2(s[2], s[3]);
    a.episode[n] = ep + 1;
    a.elapsed[n] = 0;
    a.ep_ret[n] = 0.0f;
    a.ep_len[n] = 0;
    store_obs<Env>(a.obs_out, n, obs);
}

template <class Env> __global__ void __launch_bounds__(256) k_env_step(EnvArgs a)
{
    const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (size_t)a.N) return;

    double2 *sp = reinterpret_cast<double2 *>(a.state + n * 4);
    const double2 s01 = sp[0], s23 = sp[1];
    double s[4] = {s01.x, s01.y, s23.x, s23.y};
    int elapsed = a.elapsed[n];
    float ep_ret = a.ep_ret[n];
    int ep_len = a.ep_len[n];
    const int action = a.action[n];

    float obs[6], r;
    const bool terminated = Env::step(s, action, obs, r);
    elapsed += 1;                                            // TimeLimit.step
    const bool truncated = elapsed >= MAX_EPISODE_STEPS;
    const int d = terminated ? 1 : 0;
    int tr = truncated ? 1 : 0;

    ep_ret += r;                                             // RecordEpisodeStatistics.step
    ep_len += 1;
    const bool ended = terminated || truncated;
    if (ended) {
        if (a.ret_ring) {
            const int c = a.ring_cnt[n];
            const size_t slot = n * (size_t)a.ring + (size_t)(c % a.ring);
            a.ret_ring[slot] = ep_ret;
            a.len_ring[slot] = ep_len;
            a.ring_cnt[n] = c + 1;
        }
        ep_ret = 0.0f;
        ep_len = 0;
    }

    bool can_reset = true;
    if (!ended && a.is_last) { tr += 1; can_reset = false; } // src/ppo.py:353-356
    a.reward[n] = r;
    a.done[n] = d;
    a.trunc[n] = tr;

    if (d + tr > 0) {                                        // src/ppo.py:377-382
        if (a.term_obs) store_obs<Env>(a.term_obs, n, obs);
        if (can_reset) {
            const uint32_t ep = a.episode[n];
            double u[4];
            draw_reset(a, n, ep, u);
            Env::reset(s, u, obs);
            a.episode[n] = ep + 1;
            elapsed = 0;
        }
    }
    sp[0] = make_double2(s[0], s[1]);
    sp[1] = make_double2(s[2], s[3]);
    a.elapsed[n] = elapsed;
    a.ep_ret[n] = ep_ret;
    a.ep_len[n] = ep_len;
    store_obs<Env>(a.obs_out, n, obs);
}

// ---- categorical sampling for the rollout's policy step ---------------------------------
constexpr int MAX_ACTIONS = 32;
__global__ void __launch_bounds__(256) k_categorical_sample(const float *logits, int A, uint64_t seed,
                                                            uint64_t env_base, uint64_t step, int *action,
                                                            float *logp, int N)
{
    const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (size_t)N) return;
    const float *row = logits + n * (size_t)A;
    float mx = row[0];
    for (int k = 1; k < A; ++k) mx = fmaxf(mx, row[k]);
    float sum = 0.0f;
    for (int k = 0; k < A; ++k) sum += expf(row[k] - mx);
    const float lse = mx + logf(sum);                        // log-sum-exp: Categorical(logits=...) normalisation
    const uint64_t id = env_base + n;
    // key half 1 is offset so this stream never collides with the reset stream of the same seed
    const Philox4 rnd = philox4x32_10((uint32_t)id, (uint32_t)(id >> 32), (uint32_t)step, (uint32_t)(step >> 32),
                                      (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5bd1e995u);
    const float u = (float)(rnd.v[0] >> 8) * (1.0f / 16777216.0f);  // [0,1), 24 bits
    const float target = u * sum;
    float cdf = 0.0f;
    int pick = A - 1;
    for (int k = 0; k < A; ++k) {
        cdf += expf(row[k] - mx);
        if (target < cdf) { pick = k; break; }
    }
    action[n] = pick;
    logp[n] = row[pick] - lse;
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
