/*
 * oracle/ppo_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the reference's PPO hot path, written from the
 * reference's text (BradleyGS1/PPO, /root/reference) and, for the env physics the
 * reference delegates to a third-party package, from the published algorithm of
 * gymnasium==0.29.1 (environment.yml:19-20; restated in SURVEY.md 8c).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object.  Nothing under ppo_amd/ links, imports or calls it.
 *
 * Pinning status
 *   - oracle_gae_ragged / oracle_loss:  PINNED.  Checked against the known-answer
 *     tables of the reference's tests/test_ppo.py:25-88 and :101-122 and against
 *     tests/golden/ref_*.npz, which were produced by running the reference's own
 *     PPO.compute_advantages / PPO.compute_losses (oracle/gen_golden.py).
 *   - oracle_cartpole_* / oracle_acrobot_*:  PARITY UNPINNED.  The reference holds no
 *     numeric fixture for physics and gymnasium is not installed; anchored only on
 *     hand-derived known answers.
 *   - oracle_philox4x32_10: pinned against the Random123 known-answer vectors.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fno-builtin -shared -fPIC (oracle/Makefile).
 * -fno-builtin keeps sin/cos/pow as the libm calls CPython and numpy make (gcc would
 * otherwise merge sin+cos into sincos() and turn pow(x, 2.0) into x*x, which differ from
 * them in the last bit for ~0.1% of arguments).
 * -ffp-contract=off is REQUIRED: the reference rounds after every operation
 * (torch elementwise kernels, CPython floats); a fused multiply-add would not.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_EINDEX 1 /* torch would raise IndexError */

/* ------------------------------------------------------------------------- */
/* GAE(lambda), ragged bootstrap list.  Follows src/ppo.py:452-488 line by line.
 *   rewards, values   [T,N] f32 row-major
 *   end_values        [E]   f32, actor-major (src/ppo.py:401-402)
 *   done, trunc       [T,N] i32
 * gamma/lambda arrive as Python floats (double).  `discount_factor * next_values`
 * multiplies an f32 tensor by a Python scalar => the scalar is rounded to f32;
 * `discount_factor * gae_factor * next_advantages` multiplies the two Python
 * floats in double first (left-to-right), then rounds to f32 (:482-483).        */
int oracle_gae_ragged(const float *rewards, const float *values,
                      const float *end_values, long E, const int *done,
                      const int *trunc, double gamma, double lam, float *adv,
                      int T, int N, long *scratch_idx /* [N] */)
{
    const float g = (float)gamma;
    const float gl = (float)(gamma * lam);
    /* :468-469  ep_counts = sum_t(d + tr - d*tr) ; end_indices = cumsum - 1 */
    long run = 0;
    for (int n = 0; n < N; ++n) {
        long c = 0;
        for (int t = 0; t < T; ++t) {
            int d = done[(size_t)t * N + n], tr = trunc[(size_t)t * N + n];
            c += d + tr - d * tr;
        }
        run += c;
        scratch_idx[n] = run - 1;
    }
    for (int n = 0; n < N; ++n) {
        float nv = 0.0f, na = 0.0f; /* :470-471 */
        long idx = scratch_idx[n];
        for (int t = T - 1; t >= 0; --t) { /* :473 */
            size_t o = (size_t)t * N + n;
            int d = done[o], tr = trunc[o];
            int term = d + tr;   /* :476 (may be 2) */
            long j = idx;        /* :478 gather for every column; negative wraps */
            if (j < 0) j += E;
            if (j < 0 || j >= E) return ORACLE_EINDEX;
            float ev = end_values[j];
            nv = (float)(1 - term) * nv + (float)tr * ev; /* :478 */
            na = (float)(1 - term) * na;                  /* :479 */
            idx -= term;                                  /* :480 */
            float delta = rewards[o] + g * nv - values[o]; /* :482 */
            float a = delta + gl * na;                     /* :483 */
            adv[o] = a;
            nv = values[o]; /* :485 */
            na = a;         /* :486 */
        }
    }
    return ORACLE_OK;
}

/* Dense-bootstrap form: end_value_dense[t,n] holds V(terminal obs) wherever a flag
 * is set (SURVEY.md 7 "Ragged end_values").  Same arithmetic as :476-486 with the
 * gather replaced by a same-index read.                                        */
int oracle_gae_dense(const float *rewards, const float *values,
                     const float *end_value_dense, const int *done,
                     const int *trunc, double gamma, double lam, float *adv,
                     float *ret_or_null, int T, int N)
{
    const float g = (float)gamma;
    const float gl = (float)(gamma * lam);
    for (int n = 0; n < N; ++n) {
        float nv = 0.0f, na = 0.0f;
        for (int t = T - 1; t >= 0; --t) {
            size_t o = (size_t)t * N + n;
            int d = done[o], tr = trunc[o];
            int term = d + tr;
            float ev = tr ? end_value_dense[o] : 0.0f;
            nv = (float)(1 - term) * nv + (float)tr * ev;
            na = (float)(1 - term) * na;
            float delta = rewards[o] + g * nv - values[o];
            float a = delta + gl * na;
            adv[o] = a;
            if (ret_or_null) ret_or_null[o] = a + values[o]; /* :498 returns */
            nv = values[o];
            na = a;
        }
    }
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------- */
/* Advantage statistics (src/ppo.py:501: mean, torch.std = unbiased).
 * out3 = {count, mean, M2} in double.                                          */
void oracle_adv_stats(const float *adv, long B, double *out3)
{
    double mean = 0.0;
    for (long i = 0; i < B; ++i) mean += (double)adv[i];
    mean = B > 0 ? mean / (double)B : 0.0;
    double m2 = 0.0;
    for (long i = 0; i < B; ++i) {
        double d = (double)adv[i] - mean;
        m2 += d * d;
    }
    out3[0] = (double)B;
    out3[1] = mean;
    out3[2] = m2;
}

/* Clipped-surrogate / value loss with gradients.  Follows src/ppo.py:490-527
 * (+ :595 when log-probs are given).  Elementwise work in f32 exactly as torch
 * does it, reductions in double (torch reduces in f32; see DESIGN.md tolerance).
 *   x      : prob ratios (ratio_input=1) or new log-probs (ratio_input=0)
 *   old_lp : old log-probs (ignored when ratio_input=1)
 *   stats3 : {count, mean, M2} of the advantages to normalise with
 *   out4   : policy_loss, value_loss, clip_frac, kl_div
 *   g_x    : d policy_loss / d x   (per element, already divided by B)
 *   g_v    : d value_loss  / d new_v
 * Tie rules follow torch autograd: minimum/maximum split the gradient 1/2-1/2 on
 * exact ties, clamp passes gradient on the closed interval [min,max].            */
void oracle_loss(const float *x, const float *old_lp, const float *new_v,
                 const float *old_v, const float *adv, const double *stats3,
                 double clip, int norm_adv, int clip_va, int ratio_input, long B,
                 float *out4, float *g_x, float *g_v)
{
    const float lo = (float)(1.0 - clip), hi = (float)(1.0 + clip);
    const float epsf = (float)clip;
    float mean_f = 0.0f, denom_f = 1.0f;
    if (norm_adv) {
        double n = stats3[0];
        mean_f = (float)stats3[1];
        float std_f = (float)sqrt(stats3[2] / (n - 1.0)); /* unbiased; NaN at B=1 like torch */
        denom_f = std_f + 1e-8f;                          /* :501 */
    }
    const float invB = 1.0f / (float)B;
    double s_pi = 0.0, s_v = 0.0, s_cf = 0.0, s_kl = 0.0;
    for (long i = 0; i < B; ++i) {
        float rho = ratio_input ? x[i] : expf(x[i] - old_lp[i]); /* :595 */
        float a = adv[i];
        float ret = a + old_v[i]; /* :498 from UN-normalised A */
        if (norm_adv) a = (a - mean_f) / denom_f; /* :501 */
        float rc = rho < lo ? lo : (rho > hi ? hi : rho); /* :503 */
        float w = rho * a;  /* :504 */
        float c = rc * a;   /* :505 */
        float m = w < c ? w : c; /* :508 */
        s_pi += (double)m;
        int inside = (rho >= lo) && (rho <= hi);
        float gr; /* d min / d rho */
        if (w < c) gr = a;
        else if (w > c) gr = inside ? a : 0.0f;
        else gr = 0.5f * a + (inside ? 0.5f * a : 0.0f);
        gr = -gr * invB;
        if (g_x) g_x[i] = ratio_input ? gr : gr * rho;

        float v = new_v[i];
        float e = ret - v;
        float se = e * e; /* :513 / :518 */
        float gv;
        if (clip_va) {
            float vlo = old_v[i] - epsf, vhi = old_v[i] + epsf; /* :514 */
            float vc = v < vlo ? vlo : (v > vhi ? vhi : v);
            int vin = (v >= vlo) && (v <= vhi);
            float ec = ret - vc;
            float ce = ec * ec; /* :515 */
            float mx = se > ce ? se : ce; /* :516 */
            s_v += (double)mx;
            float g_se = -2.0f * e, g_ce = vin ? -2.0f * ec : 0.0f;
            if (se > ce) gv = g_se;
            else if (se < ce) gv = g_ce;
            else gv = 0.5f * g_se + 0.5f * g_ce;
        } else {
            s_v += (double)se;
            gv = -2.0f * e;
        }
        if (g_v) g_v[i] = 0.5f * gv * invB;

        float rm1 = rho - 1.0f;
        s_cf += (fabsf(rm1) > epsf) ? 1.0 : 0.0; /* :522 */
        s_kl += (double)(rm1 - logf(rho));       /* :525 */
    }
    out4[0] = (float)(-s_pi / (double)B);
    out4[1] = (float)(0.5 * s_v / (double)B);
    out4[2] = (float)(s_cf / (double)B);
    out4[3] = (float)(s_kl / (double)B);
}

/* The same loss for a Categorical head given raw logits[B,A]: restates what
 * Agent.get_actions_and_values does on the update path with
 * torch.distributions.Categorical (src/ppo.py:186-189 logits -> distribution, :212
 * log_prob(actions), :214 entropy().mean()): normalised logits = logits -
 * logsumexp, probs = softmax, entropy = -sum p log p; then oracle_loss on the
 * gathered log-probs, and the chain rule back to the logits:
 *   d new_lp / d logit_k = [k == a] - p_k ;  d H / d logit_k = -p_k (log p_k + H).
 *   out5 = policy_loss, value_loss, clip_frac, kl_div, mean entropy                */
void oracle_loss_logits(const float *logits, int A, const int *actions, const float *old_lp,
                        const float *new_v, const float *old_v, const float *adv,
                        const double *stats3, double clip, int norm_adv, int clip_va, long B,
                        float *out5, float *g_logits_pl, float *g_logits_ent, float *g_v,
                        float *scratch_lp /* [B] */, float *scratch_glp /* [B] */)
{
    double s_h = 0.0;
    const float invB = 1.0f / (float)B;
    for (long i = 0; i < B; ++i) {
        const float *lg = logits + i * A;
        float mx = lg[0];
        for (int k = 1; k < A; ++k) mx = lg[k] > mx ? lg[k] : mx;
        float s = 0.0f;
        for (int k = 0; k < A; ++k) s += expf(lg[k] - mx);
        float lse = mx + logf(s);
        float H = 0.0f;
        for (int k = 0; k < A; ++k) {
            float lp = lg[k] - lse, p = expf(lg[k] - mx) / s;
            H -= lp * p;
        }
        scratch_lp[i] = lg[actions[i]] - lse;
        s_h += (double)H;
        if (g_logits_ent) {
            for (int k = 0; k < A; ++k) {
                float lp = lg[k] - lse, p = expf(lg[k] - mx) / s;
                g_logits_ent[i * A + k] = -p * (lp + H) * invB;
            }
        }
    }
    oracle_loss(scratch_lp, old_lp, new_v, old_v, adv, stats3, clip, norm_adv, clip_va, 0, B, out5,
                scratch_glp, g_v);
    out5[4] = (float)(s_h / (double)B);
    if (g_logits_pl) {
        for (long i = 0; i < B; ++i) {
            const float *lg = logits + i * A;
            float mx = lg[0];
            for (int k = 1; k < A; ++k) mx = lg[k] > mx ? lg[k] : mx;
            float s = 0.0f;
            for (int k = 0; k < A; ++k) s += expf(lg[k] - mx);
            for (int k = 0; k < A; ++k) {
                float p = expf(lg[k] - mx) / s;
                g_logits_pl[i * A + k] = scratch_glp[i] * ((k == actions[i] ? 1.0f : 0.0f) - p);
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Salmon et al., SC'11; Random123).  Integer arithmetic only.   */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Reset draws: 4 uniforms in [0,1) with 53 random bits each, the way
 * numpy.random.Generator.random() builds a double from two 32-bit words
 * ((a>>5)*2^26 + (b>>6)) * 2^-53.  Stream layout (ours; the reference is
 * unseeded, src/ppo.py:272,279):
 *   key = (seed_lo, seed_hi), counter = (env_lo, env_hi, episode, block)        */
void oracle_reset_uniform4(uint64_t seed, uint64_t env_id, uint32_t episode, double u[4])
{
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (uint32_t blk = 0; blk < 2; ++blk) {
        uint32_t ctr[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), episode, blk};
        uint32_t o[4];
        oracle_philox4x32_10(ctr, key, o);
        u[2 * blk + 0] = ((double)(o[0] >> 5) * 67108864.0 + (double)(o[1] >> 6)) * (1.0 / 9007199254740992.0);
        u[2 * blk + 1] = ((double)(o[2] >> 5) * 67108864.0 + (double)(o[3] >> 6)) * (1.0 / 9007199254740992.0);
    }
}

/* Action-sampling draws of one policy step for envs env_base .. env_base + N - 1 (ours; the
 * reference samples from torch's unseeded global RNG, src/ppo.py:186-189):
 *   key = (seed_lo, seed_hi ^ 0x5bd1e995), counter = (env_lo, env_hi, block_lo, block_hi) with
 *   block = step / 4: one Philox block serves four consecutive steps, word = step % 4;
 *   u = (word >> 8) * 2^-24 in [0, 1).  The sampled action is the first k with
 *   u * sum_j exp(l_j - max) < sum_{j<=k} exp(l_j - max)  (inverse CDF over the softmax).   */
void oracle_sample_uniform_batch(uint64_t seed, uint64_t env_base, uint64_t step, float *u, int N)
{
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5bd1e995u};
    for (int n = 0; n < N; ++n) {
        uint64_t id = env_base + (uint64_t)n;
        uint64_t block = step >> 2;
        uint32_t ctr[4] = {(uint32_t)id, (uint32_t)(id >> 32), (uint32_t)block, (uint32_t)(block >> 32)};
        uint32_t o[4];
        oracle_philox4x32_10(ctr, key, o);
        u[n] = (float)(o[step & 3] >> 8) * (1.0f / 16777216.0f);
    }
}

/* ------------------------------------------------------------------------- */
/* Checker for an arithmetic identity the device code relies on (not a restatement of the
 * reference): the HIP CartPole step replaces `x / total_mass` (total_mass = 0.1 + 1.0 = 1.1,
 * divisions at cartpole.py's temp / thetaacc / xacc lines, restated below) by
 *   r = RN(1 / 1.1);  q = RN(x r);  rem = x - 1.1 q (exact in an FMA);  q' = RN(q + rem r)
 * (Markstein's correction step: correctly rounded when r is the correctly rounded reciprocal and
 * q is within one ulp of the quotient).  Counts the doubles, out of n drawn with random sign,
 * random mantissa and an exponent in [e_lo, e_hi), for which q' differs from x / 1.1.          */
long oracle_div_by_total_mass_mismatches(long n, uint64_t seed, int e_lo, int e_hi)
{
    const double c = 0.1 + 1.0, r = 1.0 / c;
    uint64_t s = seed ? seed : 88172645463325252ull;
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;                      /* xorshift64 */
        const uint64_t e = (uint64_t)(1023 + e_lo) + (s >> 52) % (uint64_t)(e_hi - e_lo);
        const uint64_t bits = (s & 0x8000000000000000ull) | (e << 52) | (s & 0xFFFFFFFFFFFFFull);
        double x;
        memcpy(&x, &bits, 8);
        const double q = x * r;
        const double rem = fma(-q, c, x);
        const double q2 = fma(rem, r, q);
        bad += (q2 != x / c);
    }
    return bad;
}

/* ------------------------------------------------------------------------- */
/* CartPole-v1, gymnasium 0.29.1 classic_control/cartpole.py (step/reset) under
 * TimeLimit(max_episode_steps=500).  Reference call sites: src/env_cartpole.py:7,
 * tests/test_ppo.py:136, tests/test_vec_env.py:15,47.  PARITY UNPINNED.
 * State (x, x_dot, theta, theta_dot) is double; obs is float32 of the state.   */
#define MAX_EPISODE_STEPS 500

void oracle_cartpole_reset(double *state4, int *elapsed, float *obs4, const double u[4])
{
    for (int k = 0; k < 4; ++k) {
        state4[k] = -0.05 + (0.05 - -0.05) * u[k]; /* Generator.uniform: low + (high-low)*U */
        obs4[k] = (float)state4[k];
    }
    *elapsed = 0;
}

/* returns flags: bit0 = terminated, bit1 = truncated */
int oracle_cartpole_step(double *state4, int *elapsed, int action, float *obs4, float *reward)
{
    const double gravity = 9.8, masscart = 1.0, masspole = 0.1;
    const double total_mass = masspole + masscart;
    const double length = 0.5;
    const double polemass_length = masspole * length;
    const double force_mag = 10.0, tau = 0.02;
    const double theta_thr = 12 * 2 * M_PI / 360;
    const double x_thr = 2.4;

    double x = state4[0], x_dot = state4[1], theta = state4[2], theta_dot = state4[3];
    double force = (action == 1) ? force_mag : -force_mag;
    double costheta = cos(theta);
    double sintheta = sin(theta);
    double temp = (force + polemass_length * pow(theta_dot, 2.0) * sintheta) / total_mass;
    double thetaacc = (gravity * sintheta - costheta * temp) /
                      (length * (4.0 / 3.0 - masspole * pow(costheta, 2.0) / total_mass));
    double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
    x = x + tau * x_dot;
    x_dot = x_dot + tau * xacc;
    theta = theta + tau * theta_dot;
    theta_dot = theta_dot + tau * thetaacc;
    state4[0] = x; state4[1] = x_dot; state4[2] = theta; state4[3] = theta_dot;
    int terminated = (x < -x_thr) || (x > x_thr) || (theta < -theta_thr) || (theta > theta_thr);
    *reward = 1.0f;
    for (int k = 0; k < 4; ++k) obs4[k] = (float)state4[k];
    *elapsed += 1; /* TimeLimit.step */
    int truncated = (*elapsed >= MAX_EPISODE_STEPS);
    return terminated | (truncated << 1);
}

/* ------------------------------------------------------------------------- */
/* Acrobot-v1, gymnasium 0.29.1 classic_control/acrobot.py ("book" dynamics, rk4,
 * wrap, bound) under TimeLimit(500).  Call site: src/env_acrobot.py:7.
 * PARITY UNPINNED.  Expression association follows the Python source's
 * left-to-right evaluation; x**2 on numpy float64 scalars is libm pow(x, 2.0).  */
void acrobot_dsdt(const double s[5], double out[5])
{
    const double m1 = 1.0, m2 = 1.0, l1 = 1.0, lc1 = 0.5, lc2 = 0.5, I1 = 1.0, I2 = 1.0;
    const double g = 9.8;
    double a = s[4];
    double theta1 = s[0], theta2 = s[1], dtheta1 = s[2], dtheta2 = s[3];
    double d1 = m1 * pow(lc1, 2.0) + m2 * (pow(l1, 2.0) + pow(lc2, 2.0) + 2 * l1 * lc2 * cos(theta2)) + I1 + I2;
    double d2 = m2 * (pow(lc2, 2.0) + l1 * lc2 * cos(theta2)) + I2;
    double phi2 = m2 * lc2 * g * cos(theta1 + theta2 - M_PI / 2.0);
    double phi1 = -m2 * l1 * lc2 * pow(dtheta2, 2.0) * sin(theta2)
                  - 2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * sin(theta2)
                  + (m1 * lc1 + m2 * l1) * g * cos(theta1 - M_PI / 2)
                  + phi2;
    double ddtheta2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * pow(dtheta1, 2.0) * sin(theta2) - phi2) /
                      (m2 * pow(lc2, 2.0) + I2 - pow(d2, 2.0) / d1);
    double ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;
    out[0] = dtheta1; out[1] = dtheta2; out[2] = ddtheta1; out[3] = ddtheta2; out[4] = 0.0;
}

static double acrobot_wrap(double x, double m, double M)
{
    double diff = M - m;
    while (x > M) x = x - diff;
    while (x < m) x = x + diff;
    return x;
}

static double acrobot_bound(double x, double m, double M)
{
    double y = x > m ? x : m; /* max(x, m) */
    return y < M ? y : M;     /* min(., M) */
}

void oracle_acrobot_obs(const double *s4, float *obs6, int state_is_f32)
{
    if (state_is_f32) { /* right after reset the state array is float32 (numpy f32 ufuncs) */
        float t1 = (float)s4[0], t2 = (float)s4[1];
        obs6[0] = cosf(t1); obs6[1] = sinf(t1); obs6[2] = cosf(t2); obs6[3] = sinf(t2);
    } else {
        obs6[0] = (float)cos(s4[0]); obs6[1] = (float)sin(s4[0]);
        obs6[2] = (float)cos(s4[1]); obs6[3] = (float)sin(s4[1]);
    }
    obs6[4] = (float)s4[2]; obs6[5] = (float)s4[3];
}

void oracle_acrobot_reset(double *state4, int *elapsed, float *obs6, const double u[4])
{
    for (int k = 0; k < 4; ++k) {
        double d = -0.1 + (0.1 - -0.1) * u[k];
        state4[k] = (double)(float)d; /* .astype(np.float32) */
    }
    oracle_acrobot_obs(state4, obs6, 1);
    *elapsed = 0;
}

/* AcrobotEnv._terminal: bool(-cos(s[0]) - cos(s[1] + s[0]) > 1.0), for n states [n,4]; returns the flag of the
 * first state; flags / lhs (the left-hand side) may be NULL. */
int oracle_acrobot_terminal(const double *state, long n, int *flags, double *lhs)
{
    int first = 0;
    for (long i = 0; i < n; ++i) {
        const double *s = state + 4 * i;
        const double v = -cos(s[0]) - cos(s[1] + s[0]);
        if (lhs) lhs[i] = v;
        if (flags) flags[i] = v > 1.0;
        if (i == 0) first = v > 1.0;
    }
    return first;
}

int oracle_acrobot_step(double *state4, int *elapsed, int action, float *obs6, float *reward)
{
    const double dt = 0.2 - 0;
    const double dt2 = dt / 2.0;
    const double torque = (double)(action - 1); /* AVAIL_TORQUE = [-1., 0., +1.] */
    double y0[5] = {state4[0], state4[1], state4[2], state4[3], torque};
    double k1[5], k2[5], k3[5], k4[5], y[5];
    acrobot_dsdt(y0, k1);
    for (int i = 0; i < 5; ++i) y[i] = y0[i] + dt2 * k1[i];
    acrobot_dsdt(y, k2);
    for (int i = 0; i < 5; ++i) y[i] = y0[i] + dt2 * k2[i];
    acrobot_dsdt(y, k3);
    for (int i = 0; i < 5; ++i) y[i] = y0[i] + dt * k3[i];
    acrobot_dsdt(y, k4);
    double ns[4];
    for (int i = 0; i < 4; ++i)
        ns[i] = y0[i] + dt / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    ns[0] = acrobot_wrap(ns[0], -M_PI, M_PI);
    ns[1] = acrobot_wrap(ns[1], -M_PI, M_PI);
    ns[2] = acrobot_bound(ns[2], -4 * M_PI, 4 * M_PI);
    ns[3] = acrobot_bound(ns[3], -9 * M_PI, 9 * M_PI);
    for (int i = 0; i < 4; ++i) state4[i] = ns[i];
    int terminated = oracle_acrobot_terminal(ns, 1, NULL, NULL);
    *reward = terminated ? 0.0f : -1.0f;
    oracle_acrobot_obs(state4, obs6, 0);
    *elapsed += 1;
    int truncated = (*elapsed >= MAX_EPISODE_STEPS);
    return terminated | (truncated << 1);
}

/* ------------------------------------------------------------------------- */
/* Batched step with the rollout bookkeeping of src/ppo.py:342-398 folded in,
 * i.e. what one iteration of SyncVecEnv.rollout's t-loop does to N envs:
 *   - step every env (:294-300)
 *   - forced truncation at the last rollout step, without reset (:352-356)
 *   - terminal observation capture (:377-379) and auto-reset (:381-382)
 *   - RecordEpisodeStatistics accumulation (return/length rings of depth `ring`)
 * kind: 0 = CartPole (obs_dim 4), 1 = Acrobot (obs_dim 6).
 * reset_draw: [N,4] uniforms in [0,1) to use for resets, or NULL => Philox stream.
 * Used by the parity tests as the checker for ppo_env_step and by bench.py as
 * the single-core CPU baseline.                                                 */
void oracle_env_step_batch(int kind, double *state, int *elapsed, float *ep_ret,
                           int *ep_len, uint32_t *episode, const int *action,
                           const double *reset_draw, uint64_t seed, uint64_t env_base,
                           float *obs_out, float *term_obs, float *reward, int *done,
                           int *trunc, float *ret_ring, int *len_ring, int *ring_cnt,
                           int ring, int N, int is_last)
{
    const int od = kind == 0 ? 4 : 6;
    for (int n = 0; n < N; ++n) {
        float obs[6], r;
        int f = kind == 0 ? oracle_cartpole_step(state + 4 * n, elapsed + n, action[n], obs, &r)
                          : oracle_acrobot_step(state + 4 * n, elapsed + n, action[n], obs, &r);
        int d = f & 1, tr = (f >> 1) & 1;
        ep_ret[n] += r;
        ep_len[n] += 1;
        int ended = d | tr;
        if (ended) { /* RecordEpisodeStatistics: push on terminated or truncated */
            if (ret_ring) {
                int c = ring_cnt[n];
                ret_ring[(size_t)n * ring + (c % ring)] = ep_ret[n];
                len_ring[(size_t)n * ring + (c % ring)] = ep_len[n];
                ring_cnt[n] = c + 1;
            }
            ep_ret[n] = 0.0f;
            ep_len[n] = 0;
        }
        reward[n] = r;
        done[n] = d;
        int can_reset = 1;
        if (!ended && is_last) { tr += 1; can_reset = 0; } /* :353-356 */
        trunc[n] = tr;
        if (d + tr > 0) { /* :377-382 */
            if (term_obs) for (int k = 0; k < od; ++k) term_obs[(size_t)n * od + k] = obs[k];
            if (can_reset) {
                double u[4];
                if (reset_draw) for (int k = 0; k < 4; ++k) u[k] = reset_draw[(size_t)n * 4 + k];
                else oracle_reset_uniform4(seed, env_base + (uint64_t)n, episode[n], u);
                episode[n] += 1;
                if (kind == 0) oracle_cartpole_reset(state + 4 * n, elapsed + n, obs, u);
                else oracle_acrobot_reset(state + 4 * n, elapsed + n, obs, u);
            }
        }
        for (int k = 0; k < od; ++k) obs_out[(size_t)n * od + k] = obs[k];
    }
}

void oracle_env_reset_batch(int kind, double *state, int *elapsed, float *ep_ret,
                            int *ep_len, uint32_t *episode, const double *reset_draw,
                            uint64_t seed, uint64_t env_base, float *obs_out, int N)
{
    const int od = kind == 0 ? 4 : 6;
    for (int n = 0; n < N; ++n) {
        double u[4];
        float obs[6];
        if (reset_draw) for (int k = 0; k < 4; ++k) u[k] = reset_draw[(size_t)n * 4 + k];
        else oracle_reset_uniform4(seed, env_base + (uint64_t)n, episode[n], u);
        episode[n] += 1;
        if (kind == 0) oracle_cartpole_reset(state + 4 * n, elapsed + n, obs, u);
        else oracle_acrobot_reset(state + 4 * n, elapsed + n, obs, u);
        ep_ret[n] = 0.0f;
        ep_len[n] = 0;
        for (int k = 0; k < od; ++k) obs_out[(size_t)n * od + k] = obs[k];
    }
}


/* ---- epoch shuffle (restates ppo_amd/csrc/minibatch.hip: k_shuffle_indices; the reference's
 * np.random.shuffle(indices), src/ppo.py:575-576, is a different generator -- what is pinned here is that
 * the device permutation is the function below, that it IS a permutation, and its statistics) ---- */
void oracle_shuffle_indices(long n, uint64_t seed, uint64_t stream_id, int64_t *out)
{
    enum { ROUNDS = 6 };
    int k = 2;
    while (k < 32 && (1L << k) < n) ++k;
    const uint32_t mask = k == 32 ? 0xFFFFFFFFu : ((1u << k) - 1u);
    const int sh_a = (k + 1) / 2, sh_b = k / 3 > 1 ? k / 3 : 1;
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x2545F491u};
    uint32_t w[12];
    for (uint32_t b = 0; b < 3; ++b) {
        const uint32_t ctr[4] = {(uint32_t)stream_id, (uint32_t)(stream_id >> 32), b, 0x5348464Cu};
        oracle_philox4x32_10(ctr, key, w + 4 * b);
    }
    for (long i = 0; i < n; ++i) {
        uint32_t x = (uint32_t)i;
        do {
            for (int r = 0; r < ROUNDS; ++r) {
                x ^= x >> ((r & 1) ? sh_b : sh_a);
                x = (x * (w[r] | 1u) + w[ROUNDS + r]) & mask;
            }
            x ^= x >> sh_a;
        } while ((long)x >= n);
        out[i] = (int64_t)x;
    }
}
