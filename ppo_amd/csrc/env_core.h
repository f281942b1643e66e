// Shared device code of the env kernels: CartPole-v1 / Acrobot-v1 physics, reset draws and the
// fused per-env transition (step + TimeLimit + episode statistics + forced last-step truncation +
// terminal-observation capture + auto-reset).  Included by env.hip (one launch per time step) and
// rollout.hip (whole rollout in one launch).
#pragma once
#include "common.h"

namespace ppo_env {


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

// sin and cos for |x| < 0.3 (CartPole's pole angle: an episode ends beyond 0.2095 rad): the
// fdlibm kernel polynomials (k_sin.c / k_cos.c, error < 1 ulp) without any argument reduction --
// 25 f64 operations instead of the ~250 of the generic libm sin + cos, which are a third of the
// CartPole transition.  One rounding per operation like everything else here.
__device__ __forceinline__ void sincos_small(double x, double &sn, double &cs)
{
    const double z = x * x;
    const double rs = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06
                      + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    sn = x + (z * x) * (-1.66666666666666324348e-01 + z * rs);
    const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05
                      + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    cs = 1.0 - (0.5 * z - z * rc);
}

// sin and cos for |x| < 1.6e6 (|n| < 2^20 quarter turns): Cody-Waite reduction with pi/2 in three pieces (33 + 33 + 53
// bits, fdlibm's pio2_1 / pio2_2 / pio2_3: n pio2_1 and n pio2_2 are exact products, so the first step is exact and the
// other two round once each, relative to their own result) and the fdlibm kernel polynomials on |r| <= pi/4: ~40
// instructions and no branch, against ~70 and a large-argument branch (Payne-Hanek) in the device libm's sincos.
// Acrobot's eight calls per step are what it is for (Acrobot::trig_of); NaN in, NaN out, like libm.
__device__ __forceinline__ void sincos_bounded(double x, double &sn, double &cs)
{
    const double fn = __builtin_rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-fn, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-fn, 6.07710050630396597660e-11, r);
    r = __builtin_fma(-fn, 2.02226624871116645580e-21, r);
    // (|fn| < 2^20 for every argument this is called with, see Acrobot; beyond int's range -- only the discarded first
    // pass of a lane that step() sends through its rare branch, or a NaN state -- v_cvt_i32_f64 saturates / gives 0 and
    // the results are NaN or unused either way.  n by adding 1.5 * 2^52 and reading the low mantissa bits needs no
    // conversion at all, but measured 5 us slower per Acrobot rollout: hipcc's schedule again.)
    const int n = (int)fn;
    double s, c;
    sincos_small(r, s, c);
    const double a = (n & 1) ? c : s, b = (n & 1) ? s : c;         // quadrant: (s, c), (c, -s), (-s, -c), (-c, s)
    sn = (n & 2) ? -a : a;
    cs = ((n + 1) & 2) ? -b : b;
}

// x / 1.1 (CartPole's total_mass) as three FMAs instead of the ~12-instruction IEEE division sequence:
// q = RN(x r), r = RN(1 / 1.1); the remainder x - 1.1 q is exact in an FMA; q' = RN(q + rem r) is the
// correctly rounded quotient (Markstein's correction step with a correctly rounded reciprocal).
// Checked against x / 1.1 on 1.5e9 random doubles over 60 binades around the physics' range
// (zero mismatches), and by the single-step parity tests against the oracle, which divides.
__device__ __forceinline__ double div_by_total_mass(double x)
{
    const double c = 1.1, r = 1.0 / 1.1;
    const double q = x * r;
    const double rem = __builtin_fma(-q, c, x);
    return __builtin_fma(rem, r, q);
}

// ---- CartPole-v1 ---------------------------------------------------------------------
struct CartPole {
    static constexpr int OBS = 4;
    // The pole angle of a state that a transition left behind is inside the termination band (an episode that ends is
    // reset in the same transition), where sincos_small holds; only a state the CALLER wrote can lie outside it, and
    // only the first step after a load can meet one.  So the general sin / cos is computed where the state is loaded
    // (prime) and the step itself has no branch -- in the whole-rollout kernel a branch there splits the step's
    // scheduling region between the network and the physics (415 -> 411 us per C2 rollout without it).
    struct Aux {
        bool wide;                                 // wave-uniform: some lane's loaded angle is outside sincos_small's range
        double sn, cs;                             // sin / cos of the loaded angle, valid while `wide`
    };
    static constexpr bool PAIR_TRIG = false;       // (see Acrobot)
    static constexpr bool TWO_FORMS = false;       // (see Acrobot)
    __device__ static __forceinline__ bool wide(const double (&)[4]) { return false; }
    template <bool PAIR = false, bool GENERAL = false> __device__ static __forceinline__ void prime(const double (&s)[4], Aux &aux)
    {
        aux.wide = !__all(fabs(s[2]) < 0.3);
        aux.sn = aux.cs = 0.0;
        if (aux.wide) {                            // an env stepped past its terminal state by the caller
            aux.cs = cos(s[2]);
            aux.sn = sin(s[2]);
        }
    }

    template <bool PAIR = false>
    __device__ static __forceinline__ void reset(double (&s)[4], const double (&u)[4], float (&obs)[6], Aux &)
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[k] = -0.05 + (0.05 - -0.05) * u[k];
            obs[k] = (float)s[k];
        }
    }

    // returns terminated; reward is always 1.0
    template <bool PAIR = false, bool GENERAL = false>
    __device__ static __forceinline__ bool step(double (&s)[4], Aux &aux, int action, float (&obs)[6], float &reward)
    {
        constexpr double gravity = 9.8, masscart = 1.0, masspole = 0.1;
        constexpr double total_mass = masspole + masscart;
        const double length = 0.5;
        const double polemass_length = masspole * length;
        const double force_mag = 10.0, tau = 0.02;
        const double theta_thr = 12 * 2 * PI / 360;
        const double x_thr = 2.4;

        double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
        const double force = (action == 1) ? force_mag : -force_mag;
        double costheta, sintheta;
        sincos_small(theta, sintheta, costheta);
        sintheta = aux.wide ? aux.sn : sintheta;      // (first step after a load only, see Aux)
        costheta = aux.wide ? aux.cs : costheta;
        aux.wide = false;
        static_assert(total_mass == 1.1, "div_by_total_mass divides by 1.1");
        const double temp = div_by_total_mass(force + polemass_length * (theta_dot * theta_dot) * sintheta);
        const double thetaacc = (gravity * sintheta - costheta * temp) /
                                (length * (4.0 / 3.0 - div_by_total_mass(masspole * (costheta * costheta))));
        const double xacc = temp - div_by_total_mass(polemass_length * thetaacc * costheta);
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

    // Trigonometry is ~60 % of this env.  Per RK4 stage the book dynamics need sin / cos of theta2,
    // cos(theta1 - pi/2) and cos(theta1 + theta2 - pi/2).  The last two are sin(theta1) and
    // sin(theta1 + theta2) = sin(theta1) cos(theta2) + cos(theta1) sin(theta2): two sincos calls per
    // stage (shared argument reduction) instead of three reductions and four polynomials, and the
    // termination test's cos(theta1 + theta2) comes from the observation's own sin / cos the same
    // way: 10 calls per step instead of 15.  The identities hold to ~3e-16 absolute (the reference's
    // own arguments theta - fl(pi/2) carry that much rounding), five orders below what the RK4 stages
    // amplify an ulp to (test_single_step_from_identical_states) and eleven below the 1e-5 bar.
    // sin / cos of the state's own angles are what the observation is made of AND what the next
    // step's first RK4 stage needs: they travel with the state (Aux) in the kernels that keep an env
    // in registers across steps -- 8 calls per step; a kernel that loads the state primes them with
    // the same two calls, so both forms compute the same numbers.
    //
    // PAIR (the whole-rollout kernel's 32-env waves, small batches): lanes l and l + 32 of such a wave carry the SAME
    // env -- the matrix-core tile is 32 envs wide, the upper half-wave only holds the other halves of its operands --
    // so the two lanes split every pair of calls: the lower lane reduces theta1, the upper one theta2, and they
    // exchange the results.  One sincos per pair instead of two with all 64 lanes busy, 4 instead of 8 per step; the
    // function values are the same whichever lane computes them, so results are bit-identical to the unpaired form.
    // Every call site is reached by both lanes of a pair or by neither (they hold identical state).
    //
    // TWO_FORMS.  The calls are sincos_bounded, not libm's sincos: a state that a transition or a reset left behind has
    // |theta| <= pi, |w1| <= 4 pi, |w2| <= 9 pi (wrap and clip at the end of step()), and from such a state no RK4 stage
    // angle exceeds ~3.4e3 rad.  (With W = max |w|: |phi1| <= 1.5 W^2 + 19.6, |phi2| <= 4.9, d1 in [2.5, 4.5], d2 in
    // [0.75, 1.75], den in [0.57, 1.03], so both accelerations are below A(W) = 2.72 W^2 + 35; stage velocities
    // W1 <= 28.3 + 0.1 A(28.3) = 250, W2 <= 28.3 + 0.1 A(250) = 1.7e4, and the last stage's angles are
    // theta + 0.2 W2 <= 3.4e3 -- three orders inside sincos_bounded's range.)  Only a state the CALLER wrote can lie
    // outside those bounds; a wave that loads one (wide(), wave-uniform) runs the GENERAL form of the transition,
    // which calls libm: the whole-rollout kernel holds its step loop twice and picks one per wave at entry, so the
    // loop everybody runs carries neither the check nor the slow path.  Acrobot rollout kernel at N = 65 536, T = 128:
    // 688 -> 628 us (profiles/r03_last_session_ab.txt).
    struct Trig { double s1, c1, s2, c2; };
    using Aux = Trig;
    static constexpr bool PAIR_TRIG = true;
    static constexpr bool TWO_FORMS = true;
    __device__ static __forceinline__ bool wide(const double (&s)[4])
    {
        return !__all(fabs(s[0]) <= PI && fabs(s[1]) <= PI && fabs(s[2]) <= 4 * PI && fabs(s[3]) <= 9 * PI);
    }
    template <bool GENERAL> __device__ static __forceinline__ void sincos_of(double x, double &sn, double &cs)
    {
        if constexpr (GENERAL) sincos(x, &sn, &cs); else sincos_bounded(x, sn, cs);
    }
    template <bool PAIR = false, bool GENERAL = false> __device__ static __forceinline__ void prime(const double (&s)[4], Aux &aux)
    {
        aux = trig_of<PAIR, GENERAL>(s[0], s[1]);
    }
    template <bool PAIR = false, bool GENERAL = false> __device__ static __forceinline__ Trig trig_of(double theta1, double theta2)
    {
        Trig t;
        if constexpr (!PAIR) {
            sincos_of<GENERAL>(theta1, t.s1, t.c1);
            sincos_of<GENERAL>(theta2, t.s2, t.c2);
        } else {
            const bool upper = (threadIdx.x & 32u) != 0u;
            double sn, cs;
            sincos_of<GENERAL>(upper ? theta2 : theta1, sn, cs);
            const double so = __shfl_xor(sn, 32, 64), co = __shfl_xor(cs, 32, 64);   // the partner lane's angle
            t.s1 = upper ? so : sn; t.c1 = upper ? co : cs;
            t.s2 = upper ? sn : so; t.c2 = upper ? cs : co;
        }
        return t;
    }

    template <bool PAIR = false, bool GENERAL = false> __device__ static __forceinline__ Trig write_obs(const double (&s)[4], float (&obs)[6])
    {
        const Trig t = trig_of<PAIR, GENERAL>(s[0], s[1]);
        obs[0] = (float)t.c1; obs[1] = (float)t.s1;
        obs[2] = (float)t.c2; obs[3] = (float)t.s2;
        obs[4] = (float)s[2]; obs[5] = (float)s[3];
        return t;
    }

    template <bool PAIR = false>
    __device__ static __forceinline__ void reset(double (&s)[4], const double (&u)[4], float (&obs)[6], Aux &aux)
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = (double)(float)(-0.1 + (0.1 - -0.1) * u[k]);  // .astype(float32)
        aux = write_obs<PAIR>(s, obs);
    }

    // 1 / d for d in [0.5, 5] (the two denominators of dsdt: d1 = 3.5 + cos(theta2) and
    // 1.25 - d2^2 / d1 in [0.57, 1.03]): the hardware estimate and two Newton steps, without the range
    // scaling and fix-up instructions of the general IEEE sequence.  quot() then rounds n / d like the
    // division does (Markstein's correction with the refined reciprocal): a stage's four divisions,
    // three of them by d1, cost two reciprocals and four 3-instruction quotients.
    __device__ static __forceinline__ double recip(double d)
    {
        double r = __builtin_amdgcn_rcp(d);
        r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
        r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
        return r;
    }
    __device__ static __forceinline__ double quot(double n, double d, double r)
    {
        const double q = n * r;
        return __builtin_fma(__builtin_fma(-q, d, n), r, q);
    }

    __device__ static __forceinline__ void dsdt(const double (&y)[4], const Trig &t, double a, double (&out)[4])
    {
        const double m1 = 1.0, m2 = 1.0, l1 = 1.0, lc1 = 0.5, lc2 = 0.5, I1 = 1.0, I2 = 1.0, g = 9.8;
        const double dtheta1 = y[2], dtheta2 = y[3];
        const double c2 = t.c2, s2 = t.s2;
        const double d1 = m1 * (lc1 * lc1) + m2 * ((l1 * l1) + (lc2 * lc2) + 2 * l1 * lc2 * c2) + I1 + I2;
        const double d2 = m2 * ((lc2 * lc2) + l1 * lc2 * c2) + I2;
        const double r1 = recip(d1);
        const double phi2 = m2 * lc2 * g * __builtin_fma(t.s1, c2, t.c1 * s2);   // cos(theta1 + theta2 - pi/2)
        const double phi1 = -m2 * l1 * lc2 * (dtheta2 * dtheta2) * s2
                            - 2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * s2
                            + (m1 * lc1 + m2 * l1) * g * t.s1                       // cos(theta1 - pi/2)
                            + phi2;
        const double den = m2 * (lc2 * lc2) + I2 - quot(d2 * d2, d1, r1);
        const double ddtheta2 = quot(a + quot(d2, d1, r1) * phi1 - m2 * l1 * lc2 * (dtheta1 * dtheta1) * s2 - phi2,
                                     den, recip(den));
        const double ddtheta1 = quot(-(d2 * ddtheta2 + phi1), d1, r1);
        out[0] = dtheta1; out[1] = dtheta2; out[2] = ddtheta1; out[3] = ddtheta2;
    }

    __device__ static __forceinline__ double wrap(double x, double m, double M)
    {
        const double diff = M - m;
        while (x > M) x = x - diff;
        while (x < m) x = x + diff;
        return x;
    }

    template <bool PAIR = false, bool GENERAL = false>
    __device__ static __forceinline__ bool step(double (&s)[4], Aux &aux, int action, float (&obs)[6], float &reward)
    {
        const double dt = 0.2 - 0;
        const double dt2 = dt / 2.0;
        const double torque = (double)(action - 1);
        double k1[4], k2[4], k3[4], k4[4], y[4];
        dsdt(s, aux, torque, k1);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = s[i] + dt2 * k1[i];
        dsdt(y, trig_of<PAIR, GENERAL>(y[0], y[1]), torque, k2);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = s[i] + dt2 * k2[i];
        dsdt(y, trig_of<PAIR, GENERAL>(y[0], y[1]), torque, k3);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = s[i] + dt * k3[i];
        dsdt(y, trig_of<PAIR, GENERAL>(y[0], y[1]), torque, k4);
        double ns[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ns[i] = s[i] + dt / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
        // wrap(): the first trip of each of the reference's two loops as selects -- all there is to do for |x| <= 3 pi;
        // whatever is left (a lane still outside) is finished in the step's one rare branch below
        ns[0] = wrap_once(ns[0]);
        ns[1] = wrap_once(ns[1]);
        const bool unwrapped = fabs(ns[0]) > PI || fabs(ns[1]) > PI;
        ns[2] = fmin(fmax(ns[2], -4 * PI), 4 * PI);
        ns[3] = fmin(fmax(ns[3], -9 * PI), 9 * PI);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = ns[i];
        Trig t = aux = write_obs<PAIR, GENERAL>(s, obs);
        // gymnasium's termination test, bool(-cos(s[0]) - cos(s[1] + s[0]) > 1.0).  cos(theta1 + theta2) from the
        // observation's own sin / cos (angle-sum identity) saves a trigonometric reduction per step and agrees with
        // the direct value to a few 1e-16 -- so the identity DECIDES only where it cannot be wrong: outside a band of
        // 1e-9 around the threshold, a million times its worst-case error.  Inside the band the reference's own
        // expression is evaluated with the library's cos -- of theta1 and of the summed angle -- so the flag is,
        // structurally, that expression's.
        // ONE rare branch per step (wave-uniform; about one step in 10^9 takes it inside a rollout) holds both that and
        // the rest of a long wrap(): every such branch splits the step's scheduling region whether taken or not -- the
        // band's cost 8 us of a 701 us rollout at N = 65536 (profiles/r03_acrobot_guard_band_ab.txt), wrap()'s four
        // while loops 12 us of 626 (profiles/r03_last_session_ab.txt).
        double lhs = term_lhs(t);
        bool terminated = lhs > 1.0;
        if (__any(unwrapped || in_band(lhs))) {
            if (unwrapped) {                          // (both lanes of a PAIR hold the same state: they come here together)
                s[0] = wrap(s[0], -PI, PI);
                s[1] = wrap(s[1], -PI, PI);
                t = aux = write_obs<PAIR, GENERAL>(s, obs);
                lhs = term_lhs(t);
                terminated = lhs > 1.0;
            }
            if (in_band(lhs)) terminated = terminal_exact(s);
        }
        reward = terminated ? 0.0f : -1.0f;
        return terminated;
    }
    __device__ static __forceinline__ double wrap_once(double x)
    {
        x = x > PI ? x - (PI - -PI) : x;
        x = x < -PI ? x + (PI - -PI) : x;
        return x;
    }
    __device__ static __forceinline__ double term_lhs(const Trig &t) { return -t.c1 - __builtin_fma(t.c1, t.c2, -(t.s1 * t.s2)); }
    __device__ static __forceinline__ bool in_band(double lhs)
    {
#ifndef PPO_ACROBOT_NO_GUARD_BAND                     // (tools/build_variant.sh: the A/B build that prices the band)
        return fabs(lhs - 1.0) < 1e-9;
#else
        return false;
#endif
    }
    __device__ static __forceinline__ bool terminal_exact(const double (&s)[4]) { return -cos(s[0]) - cos(s[1] + s[0]) > 1.0; }
    // the decision on its own, as step() takes it (test hook k_test_acrobot_terminal)
    __device__ static __forceinline__ bool terminal(const double (&s)[4], const Trig &t)
    {
        const double lhs = term_lhs(t);
        bool terminated = lhs > 1.0;
        if (__any(in_band(lhs))) {
            if (in_band(lhs)) terminated = terminal_exact(s);
        }
        return terminated;
    }

};

// Streaming variant: non-temporal stores for rollout buffers that nothing on the device reads again
// soon, so that they do not push the buffers the GAE kernel reads next out of the Infinity Cache.
template <class Env> __device__ __forceinline__ void store_obs_stream(float *dst, size_t n, const float (&obs)[6])
{
    using u4 = __attribute__((ext_vector_type(4))) float;
    using u2 = __attribute__((ext_vector_type(2))) float;
    if (Env::OBS == 4) {
        const u4 v = {obs[0], obs[1], obs[2], obs[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<u4 *>(dst + n * 4));
    } else {
        u2 *p = reinterpret_cast<u2 *>(dst + n * 6);
        const u2 a = {obs[0], obs[1]}, b = {obs[2], obs[3]}, c = {obs[4], obs[5]};
        __builtin_nontemporal_store(a, p);
        __builtin_nontemporal_store(b, p + 1);
        __builtin_nontemporal_store(c, p + 2);
    }
}

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

// DRAWS = false compiles the caller-supplied reset stream out: a global load anywhere in the step
// makes the compiler wait on vmcnt, which also counts the wave's stores still in flight.
template <bool DRAWS = true>
__device__ __forceinline__ void draw_reset(const EnvArgs &a, size_t n, uint32_t episode, double (&u)[4])
{
    if (DRAWS && a.reset_draw) {
        const double2 *p = reinterpret_cast<const double2 *>(a.reset_draw + n * 4);
        const double2 lo = p[0], hi = p[1];
        u[0] = lo.x; u[1] = lo.y; u[2] = hi.x; u[3] = hi.y;
    } else {
        reset_uniform4(a.seed, a.env_base + n, episode, u);
    }
}


// Per-env registers carried across steps by the kernels.
struct EnvRegs {
    double s[4];
    int elapsed;
    float ep_ret;
    int ep_len;
    int ring_cnt;      // finished episodes of this env so far (next ring slot = ring_cnt % ring)
    uint32_t episode;  // reset counter = Philox position of the next reset draw
};

struct StepOut {
    float obs[6];      // next observation (already the reset observation if the env was reset)
    float term[6];     // terminal observation, valid when done + trunc > 0
    float reward;
    int done, trunc;
};

// A reset draw computed ahead of its use (whole-rollout kernel only).  The draw of env n's next
// reset is a pure function of (seed, env id, e.episode), so it can be made at any time before that
// reset.  With a random policy some env of a wave's 64 ends an episode on ~95 % of the steps, and
// the ~200 instructions of the two Philox calls would run for the whole wave each time; instead
// every lane without a draw makes one whenever some lane needs one it does not have, which is
// about every ninth step (a lane that has just used its draw cannot need the next one earlier
// than its new episode ends).
struct ResetAhead {
    double u[4];
    bool have = false;
};

// One transition of env n.  Follows gymnasium env.step -> TimeLimit -> RecordEpisodeStatistics and
// the actor loop of the reference's rollout (src/ppo.py:342-388).
// PAIR: the caller runs this for BOTH lanes that carry env n in a 32-env wave (Acrobot::trig_of); `owner` is false in
// the lane that only helps -- it keeps its copy of the env in step but writes nothing to memory.
template <class Env, bool DRAWS = true, bool PAIR = false, bool GENERAL = false>
__device__ __forceinline__ void env_transition(EnvRegs &e, typename Env::Aux &aux, int action, const EnvArgs &a, size_t n, int is_last,
                                               StepOut &o, ResetAhead *ahead = nullptr, bool owner = true)
{
    const bool terminated = Env::template step<PAIR, GENERAL>(e.s, aux, action, o.obs, o.reward);
    e.elapsed += 1;                                          // TimeLimit.step
    const bool truncated = e.elapsed >= MAX_EPISODE_STEPS;
    o.done = terminated ? 1 : 0;
    o.trunc = truncated ? 1 : 0;

    e.ep_ret += o.reward;                                    // RecordEpisodeStatistics.step
    e.ep_len += 1;
    const bool ended = terminated || truncated;
    if (ended) {
        if (a.ret_ring && (!PAIR || owner)) {
            // counters live in registers: a load here would make the wave wait for every store
            // it has in flight (vmcnt counts both), once per step in practice since some env of
            // the 64 ends an episode almost every step
            const size_t slot = n * (size_t)a.ring + (size_t)(e.ring_cnt % a.ring);
            a.ret_ring[slot] = e.ep_ret;
            a.len_ring[slot] = e.ep_len;
        }
        if (a.ret_ring) e.ring_cnt += 1;
        e.ep_ret = 0.0f;
        e.ep_len = 0;
    }
    bool can_reset = true;
    if (!ended && is_last) { o.trunc += 1; can_reset = false; }  // src/ppo.py:353-356
    const bool flagged = o.done + o.trunc > 0;
    if (ahead) {                                             // compile-time: the caller passes it or not
        if (__any(flagged && can_reset && !ahead->have)) {
            if (!ahead->have) {
                reset_uniform4(a.seed, a.env_base + n, e.episode, ahead->u);
                ahead->have = true;
            }
        }
    }
    if (flagged) {                                           // src/ppo.py:377-382
#pragma unroll
        for (int k = 0; k < 6; ++k) o.term[k] = o.obs[k];
        if (can_reset) {
            if (ahead) {
                Env::template reset<PAIR>(e.s, ahead->u, o.obs, aux);
                ahead->have = false;
            } else {
                double u[4];
                draw_reset<DRAWS>(a, n, e.episode, u);
                Env::template reset<PAIR>(e.s, u, o.obs, aux);
            }
            e.episode += 1;
            e.elapsed = 0;
        }
    }
}

__device__ __forceinline__ void load_env(EnvRegs &e, const EnvArgs &a, size_t n)
{
    const double2 *sp = reinterpret_cast<const double2 *>(a.state + n * 4);
    const double2 s01 = sp[0], s23 = sp[1];
    e.s[0] = s01.x; e.s[1] = s01.y; e.s[2] = s23.x; e.s[3] = s23.y;
    e.elapsed = a.elapsed[n];
    e.ep_ret = a.ep_ret[n];
    e.ep_len = a.ep_len[n];
    e.ring_cnt = a.ring_cnt ? a.ring_cnt[n] : 0;
    e.episode = a.episode[n];
}

__device__ __forceinline__ void store_env(const EnvRegs &e, const EnvArgs &a, size_t n)
{
    double2 *sp = reinterpret_cast<double2 *>(a.state + n * 4);
    sp[0] = make_double2(e.s[0], e.s[1]);
    sp[1] = make_double2(e.s[2], e.s[3]);
    a.elapsed[n] = e.elapsed;
    a.ep_ret[n] = e.ep_ret;
    a.ep_len[n] = e.ep_len;
    if (a.ring_cnt) a.ring_cnt[n] = e.ring_cnt;
    a.episode[n] = e.episode;
}

// Categorical(logits) sample + log-prob for env `id` at policy step `step` (reference
// src/ppo.py:186-189,212).  Key half 1 is offset so this stream never collides with the reset
// stream of the same seed.
// The uniform draw of (seed; env id, policy step): it depends on nothing the network computes, so the
// whole-rollout kernel makes it before the forward pass, where its ten dependent Philox rounds fill
// VALU slots in the shadow of the first MFMAs instead of sitting between the logits and the action.
// One Philox4x32-10 block serves FOUR consecutive policy steps of an env: block = step / 4 (the
// counter), word = step % 4.  The whole-rollout kernel therefore runs the ten rounds on every fourth
// step only (SampleDraws below); this function is the definition, used by the per-step kernel.
__device__ __forceinline__ Philox4 sample_block(uint64_t seed, uint64_t id, uint64_t block)
{
    return philox4x32_10((uint32_t)id, (uint32_t)(id >> 32), (uint32_t)block, (uint32_t)(block >> 32),
                         (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5bd1e995u);
}
__device__ __forceinline__ float uniform24(uint32_t word) { return (float)(word >> 8) * (1.0f / 16777216.0f); }  // [0,1)

__device__ __forceinline__ float sample_uniform(uint64_t seed, uint64_t id, uint64_t step)
{
    const Philox4 rnd = sample_block(seed, id, step >> 2);
    const uint32_t w = (uint32_t)step & 3u;
    return uniform24(w == 0 ? rnd.v[0] : (w == 1 ? rnd.v[1] : (w == 2 ? rnd.v[2] : rnd.v[3])));
}

// The same stream, one step after the other: next(step) == sample_uniform(seed, id, step) for
// consecutive steps.  Holds the unused words of the current block; the ten rounds run when they run
// out (every fourth step, at a wave-uniform point since all lanes share `step`).
struct SampleDraws {
    uint32_t w0, w1, w2;   // words still to hand out, in order
    __device__ __forceinline__ float next(uint64_t seed, uint64_t id, uint64_t step, bool first)
    {
        const uint32_t k = (uint32_t)step & 3u;
        uint32_t word;
        if (k == 0 || first) {                       // wave-uniform
            const Philox4 rnd = sample_block(seed, id, step >> 2);
            word = k == 0 ? rnd.v[0] : (k == 1 ? rnd.v[1] : (k == 2 ? rnd.v[2] : rnd.v[3]));
            w0 = k == 0 ? rnd.v[1] : (k == 1 ? rnd.v[2] : rnd.v[3]);
            w1 = k == 0 ? rnd.v[2] : rnd.v[3];
            w2 = rnd.v[3];
        } else {
            word = w0; w0 = w1; w1 = w2;
        }
        return uniform24(word);
    }
};

__device__ __forceinline__ void categorical_from_uniform(const float *lg, int A, float u, int &action, float &logp)
{
    float mx = lg[0];
    for (int k = 1; k < A; ++k) mx = fmaxf(mx, lg[k]);
    float sum = 0.0f;
    for (int k = 0; k < A; ++k) sum += expf(lg[k] - mx);
    const float lse = mx + logf(sum);
    const float target = u * sum;
    float cdf = 0.0f;
    int pick = A - 1;
    float lpick = lg[A - 1];
    bool found = false;
    for (int k = 0; k < A; ++k) {
        cdf += expf(lg[k] - mx);
        if (!found && target < cdf) { pick = k; lpick = lg[k]; found = true; }
    }
    action = pick;
    logp = lpick - lse;
}

__device__ __forceinline__ void categorical_sample(const float *lg, int A, uint64_t seed, uint64_t id, uint64_t step,
                                                   int &action, float &logp)
{
    categorical_from_uniform(lg, A, sample_uniform(seed, id, step), action, logp);
}

}  // namespace ppo_env
