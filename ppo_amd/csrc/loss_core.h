// Per-sample PPO loss terms shared by the loss kernels (loss.hip) and the fused dense update
// (update.hip): the reference's compute_losses (src/ppo.py:490-527) and the Categorical front end
// (:201-216), elementwise in f32, one rounding per operation.
#pragma once
#include "common.h"

namespace ppo_loss {

// Per-sample terms shared by both front ends (ratio / log-prob input, and logits input).
struct LossConsts {
    float lo, hi, epsf, mean_f, denom_f, invB;
    int norm_adv, clip_va;
};

struct SampleOut {
    float m, vterm, clipped, kl;  // min(w, c), value term, |rho-1| > eps, (rho-1) - log rho
    float g_rho, g_v;             // d policy_loss / d rho, d value_loss / d new_v (both already / B)
};

__device__ __forceinline__ LossConsts loss_consts(const double *stats3, int norm_adv, int clip_va, float lo,
                                                  float hi, float epsf, long B)
{
    LossConsts c;
    c.lo = lo; c.hi = hi; c.epsf = epsf; c.norm_adv = norm_adv; c.clip_va = clip_va;
    c.mean_f = 0.0f; c.denom_f = 1.0f;
    if (norm_adv) {
        const double n = stats3[0];
        c.mean_f = (float)stats3[1];
        const float std_f = (float)sqrt(stats3[2] / (n - 1.0));  // unbiased (torch.std), NaN at B=1
        c.denom_f = std_f + 1e-8f;                               // src/ppo.py:501
    }
    c.invB = 1.0f / (float)B;
    return c;
}

__device__ __forceinline__ SampleOut loss_sample(const LossConsts &c, float rho, float adv, float ov, float v)
{
    SampleOut o;
    const float ret = adv + ov;                                       // :498
    if (c.norm_adv) adv = (adv - c.mean_f) / c.denom_f;               // :501
    const float rc = rho < c.lo ? c.lo : (rho > c.hi ? c.hi : rho);   // :503
    const float w = rho * adv, cl = rc * adv;                         // :504-505
    o.m = w < cl ? w : cl;                                            // :508
    const bool inside = (rho >= c.lo) && (rho <= c.hi);
    float gr;
    if (w < cl) gr = adv;
    else if (w > cl) gr = inside ? adv : 0.0f;
    else gr = 0.5f * adv + (inside ? 0.5f * adv : 0.0f);              // torch.minimum tie
    o.g_rho = -gr * c.invB;

    const float e = ret - v;
    const float se = e * e;                                           // :513 / :518
    float gvk;
    if (c.clip_va) {
        const float vlo = ov - c.epsf, vhi = ov + c.epsf;             // :514
        const float vc = v < vlo ? vlo : (v > vhi ? vhi : v);
        const bool vin = (v >= vlo) && (v <= vhi);
        const float ec = ret - vc;
        const float ce = ec * ec;                                     // :515
        o.vterm = se > ce ? se : ce;                                  // :516
        const float g_se = -2.0f * e, g_ce = vin ? -2.0f * ec : 0.0f;
        if (se > ce) gvk = g_se;
        else if (se < ce) gvk = g_ce;
        else gvk = 0.5f * g_se + 0.5f * g_ce;                         // torch.maximum tie
    } else {
        o.vterm = se;
        gvk = -2.0f * e;
    }
    o.g_v = 0.5f * gvk * c.invB;
    const float rm1 = rho - 1.0f;
    o.clipped = (fabsf(rm1) > c.epsf) ? 1.0f : 0.0f;                  // :522
    o.kl = rm1 - logf(rho);                                           // :525
    return o;
}


// Categorical front end for one sample: normalised log-probs, probs, entropy, log-prob of the taken
// action (NaN for an action outside [0, A): torch's Categorical.log_prob raises).
template <int A> struct Categorical {
    float lp[A], pr[A], H, new_lp;
    __device__ __forceinline__ void from_logits(const float (&lg)[A], int act)
    {
        float mx = lg[0];
#pragma unroll
        for (int k = 1; k < A; ++k) mx = fmaxf(mx, lg[k]);
        float ex[A], s = 0.0f;
#pragma unroll
        for (int k = 0; k < A; ++k) { ex[k] = expf(lg[k] - mx); s += ex[k]; }
        const float lse = mx + logf(s);                 // torch.logsumexp
        H = 0.0f; new_lp = 0.0f;
#pragma unroll
        for (int k = 0; k < A; ++k) {
            lp[k] = lg[k] - lse;                        // Categorical.logits (normalised)
            pr[k] = ex[k] / s;                          // Categorical.probs (softmax)
            H -= lp[k] * pr[k];                         // Categorical.entropy
            if (k == act) new_lp = lp[k];               // Categorical.log_prob
        }
        if ((unsigned)act >= (unsigned)A) new_lp = __builtin_nanf("");
    }
};

}  // namespace ppo_loss
