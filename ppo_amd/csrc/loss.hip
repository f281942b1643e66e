// Fused PPO minibatch loss (forward + backward) and advantage statistics -- gfx950.
//
// Replaces the ratio at reference src/ppo.py:595 and PPO.compute_losses (:490-527) plus
// the autograd backward of policy_loss and value_loss.  HBM-bound: 20 B/sample read
// (+4 for the statistics pre-pass), 8 B/sample of gradients written.  No MFMA.
//
// Structure: grid-stride over float4 groups (16 B per lane per array, the coalescing
// sweet spot), per-lane fp64 partial sums, wave64 butterfly (DPP) reduction, one LDS
// stage across the block's waves, one fp64 partial row per block in the caller's
// workspace, then a one-block finalize kernel that adds the rows in index order.  The
// grid depends only on B, so results are run-to-run deterministic.
//
// Elementwise arithmetic is the reference's f32 sequence, one rounding per operation
// (-ffp-contract=off); only the reductions differ (fp64 here, f32 tree in torch).
#include "common.h"
#include "loss_core.h"

using namespace ppo_loss;

namespace {

constexpr int LOSS_BLOCK = 256;
constexpr int LOSS_MAX_BLOCKS = 2048;  // 256 CUs x 8 blocks (guide: Guideline 11)

__host__ __device__ inline long loss_groups(long B) { return (B + 3) / 4; }

inline int loss_grid(long B)
{
    long blocks = (loss_groups(B) + LOSS_BLOCK - 1) / LOSS_BLOCK;
    if (blocks < 1) blocks = 1;
    if (blocks > LOSS_MAX_BLOCKS) blocks = LOSS_MAX_BLOCKS;
    return (int)blocks;
}

// Sum NV doubles per lane over the block; lane 0 of wave 0 writes them to dst[0..NV).
template <int NV> __device__ __forceinline__ void block_reduce_store(double (&acc)[NV], double *dst)
{
    __shared__ double part[LOSS_BLOCK / PPO_WAVE][NV];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) part[wid][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double s = 0.0;
            for (int w = 0; w < LOSS_BLOCK / PPO_WAVE; ++w) s += part[w][k];
            dst[k] = s;
        }
    }
}

__device__ __forceinline__ float4 ld4(const float *p, long i, long B, bool vec)
{
    if (vec && i + 3 < B) return *reinterpret_cast<const float4 *>(p + i);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i + 0 < B) v.x = p[i + 0];
    if (i + 1 < B) v.y = p[i + 1];
    if (i + 2 < B) v.z = p[i + 2];
    if (i + 3 < B) v.w = p[i + 3];
    return v;
}

__device__ __forceinline__ void st4(float *p, long i, long B, bool vec, const float4 &v)
{
    if (vec && i + 3 < B) { *reinterpret_cast<float4 *>(p + i) = v; return; }
    if (i + 0 < B) p[i + 0] = v.x;
    if (i + 1 < B) p[i + 1] = v.y;
    if (i + 2 < B) p[i + 2] = v.z;
    if (i + 3 < B) p[i + 3] = v.w;
}

// ---- advantage statistics: shifted sums, K = adv[0] -------------------------------
__global__ void __launch_bounds__(LOSS_BLOCK) k_adv_stats_partial(const float *adv, long B, double *partials, bool vec)
{
    const float K = adv[0];
    double acc[2] = {0.0, 0.0};
    const long G = loss_groups(B);
    for (long g = (long)blockIdx.x * LOSS_BLOCK + threadIdx.x; g < G; g += (long)gridDim.x * LOSS_BLOCK) {
        const long i = g * 4;
        const float4 a = ld4(adv, i, B, vec);
        const float e[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i + k < B) {
                const double d = (double)e[k] - (double)K;
                acc[0] += d;
                acc[1] += d * d;
            }
        }
    }
    block_reduce_store<2>(acc, partials + (size_t)blockIdx.x * 2);
}

// The same sums over adv[idx[i]], i < B: thread, group and order of additions as above, so that the statistics of a
// minibatch given by its indices equal those of the gathered copy bit for bit.
__global__ void __launch_bounds__(LOSS_BLOCK) k_adv_stats_partial_idx(const float *adv, const long *idx, long B, double *partials)
{
    const float K = adv[idx[0]];
    double acc[2] = {0.0, 0.0};
    const long G = loss_groups(B);
    for (long g = (long)blockIdx.x * LOSS_BLOCK + threadIdx.x; g < G; g += (long)gridDim.x * LOSS_BLOCK) {
        const long i = g * 4;
        float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k < B) e[k] = adv[idx[i + k]];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i + k < B) {
                const double d = (double)e[k] - (double)K;
                acc[0] += d;
                acc[1] += d * d;
            }
        }
    }
    block_reduce_store<2>(acc, partials + (size_t)blockIdx.x * 2);
}

__global__ void __launch_bounds__(LOSS_BLOCK) k_adv_stats_final(const float *adv, const long *idx, long B, const double *partials,
                                                              int nblocks, double *stats3)
{
    double acc[2] = {0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += LOSS_BLOCK) {
        acc[0] += partials[(size_t)b * 2 + 0];
        acc[1] += partials[(size_t)b * 2 + 1];
    }
    __shared__ double tot[2];
    block_reduce_store<2>(acc, tot);
    if (threadIdx.x == 0) {
        const double n = (double)B, K = (double)adv[idx ? idx[0] : 0];
        const double S = tot[0], Q = tot[1];
        stats3[0] = n;
        stats3[1] = K + S / n;
        double m2 = Q - S * S / n;
        stats3[2] = m2 > 0.0 ? m2 : 0.0;
    }
}

// ---- loss forward + backward ---------------------------------------------------------
struct LossArgs {
    const float *x, *old_lp, *new_v, *old_v, *adv;
    const double *stats3;
    float *g_x, *g_v;
    double *partials;
    float lo, hi, epsf;
    long B;
    int norm_adv, clip_va, ratio_input;
    bool vec;
};

__global__ void __launch_bounds__(LOSS_BLOCK) k_loss_partial(LossArgs a)
{
    const LossConsts c = loss_consts(a.stats3, a.norm_adv, a.clip_va, a.lo, a.hi, a.epsf, a.B);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const long G = loss_groups(a.B);
    for (long g = (long)blockIdx.x * LOSS_BLOCK + threadIdx.x; g < G; g += (long)gridDim.x * LOSS_BLOCK) {
        const long i = g * 4;
        const float4 x4 = ld4(a.x, i, a.B, a.vec);
        const float4 nv4 = ld4(a.new_v, i, a.B, a.vec);
        const float4 ov4 = ld4(a.old_v, i, a.B, a.vec);
        const float4 ad4 = ld4(a.adv, i, a.B, a.vec);
        float4 ol4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!a.ratio_input) ol4 = ld4(a.old_lp, i, a.B, a.vec);
        const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, nvs[4] = {nv4.x, nv4.y, nv4.z, nv4.w};
        const float ovs[4] = {ov4.x, ov4.y, ov4.z, ov4.w}, ads[4] = {ad4.x, ad4.y, ad4.z, ad4.w};
        const float ols[4] = {ol4.x, ol4.y, ol4.z, ol4.w};
        float gx[4], gv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float rho = a.ratio_input ? xs[k] : expf(xs[k] - ols[k]);  // :595
            const SampleOut o = loss_sample(c, rho, ads[k], ovs[k], nvs[k]);
            gx[k] = a.ratio_input ? o.g_rho : o.g_rho * rho;
            gv[k] = o.g_v;
            if (i + k < a.B) {
                acc[0] += (double)o.m;
                acc[1] += (double)o.vterm;
                acc[2] += (double)o.clipped;
                acc[3] += (double)o.kl;
            }
        }
        if (a.g_x) st4(a.g_x, i, a.B, a.vec, make_float4(gx[0], gx[1], gx[2], gx[3]));
        if (a.g_v) st4(a.g_v, i, a.B, a.vec, make_float4(gv[0], gv[1], gv[2], gv[3]));
    }
    block_reduce_store<4>(acc, a.partials + (size_t)blockIdx.x * 4);
}

// ---- discrete-policy front end: logits in, entropy and both logits gradients out ----------
// Folds what the reference does with torch.distributions.Categorical in get_actions_and_values
// (src/ppo.py:201-216: logits - logsumexp, log_prob gather, entropy = -sum p log p, .mean()) into
// the loss pass, so the update never materialises log-softmax / probs / entropy tensors.
struct LossLogitsArgs {
    const float *logits;   // [B, A] row-major
    const int *actions;    // [B]
    const float *old_lp, *new_v, *old_v, *adv;
    const double *stats3;
    float *g_logits_pl, *g_logits_ent, *g_v;  // [B, A], [B, A], [B]; any may be NULL
    double *partials;
    float lo, hi, epsf;
    long B;
    int norm_adv, clip_va;
};

template <int A> __global__ void __launch_bounds__(LOSS_BLOCK) k_loss_logits_partial(LossLogitsArgs a)
{
    const LossConsts c = loss_consts(a.stats3, a.norm_adv, a.clip_va, a.lo, a.hi, a.epsf, a.B);
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (long i = (long)blockIdx.x * LOSS_BLOCK + threadIdx.x; i < a.B; i += (long)gridDim.x * LOSS_BLOCK) {
        float lg[A];
#pragma unroll
        for (int k = 0; k < A; ++k) lg[k] = a.logits[i * A + k];
        const int act = a.actions[i];
        // an action outside [0, A) has no log-prob: torch's Categorical.log_prob raises; here the
        // sample poisons every output (NaN losses and gradients) instead of passing as new_lp = 0
        Categorical<A> cat;
        cat.from_logits(lg, act);
        const float (&lp)[A] = cat.lp;
        const float (&pr)[A] = cat.pr;
        const float H = cat.H, new_lp = cat.new_lp;
        const float rho = expf(new_lp - a.old_lp[i]);   // :595
        const SampleOut o = loss_sample(c, rho, a.adv[i], a.old_v[i], a.new_v[i]);
        const float g_lp = o.g_rho * rho;               // d policy_loss / d new_lp
        if (a.g_logits_pl) {
#pragma unroll
            for (int k = 0; k < A; ++k) a.g_logits_pl[i * A + k] = g_lp * ((k == act ? 1.0f : 0.0f) - pr[k]);
        }
        if (a.g_logits_ent) {                           // d mean(H) / d logits_k = -p_k (log p_k + H) / B
#pragma unroll
            for (int k = 0; k < A; ++k) a.g_logits_ent[i * A + k] = -pr[k] * (lp[k] + H) * c.invB;
        }
        if (a.g_v) a.g_v[i] = o.g_v;
        acc[0] += (double)o.m;
        acc[1] += (double)o.vterm;
        acc[2] += (double)o.clipped;
        acc[3] += (double)o.kl;
        acc[4] += (double)H;
    }
    block_reduce_store<5>(acc, a.partials + (size_t)blockIdx.x * 5);
}

__global__ void __launch_bounds__(LOSS_BLOCK) k_loss_logits_final(const double *partials, int nblocks, long B,
                                                                float *out5)
{
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += LOSS_BLOCK) {
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k] += partials[(size_t)b * 5 + k];
    }
    __shared__ double tot[5];
    block_reduce_store<5>(acc, tot);
    if (threadIdx.x == 0) {
        const double n = (double)B;
        out5[0] = (float)(-tot[0] / n);
        out5[1] = (float)(0.5 * tot[1] / n);
        out5[2] = (float)(tot[2] / n);
        out5[3] = (float)(tot[3] / n);
        out5[4] = (float)(tot[4] / n);
    }
}

__global__ void __launch_bounds__(LOSS_BLOCK) k_loss_final(const double *partials, int nblocks, long B, float *out4)
{
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += LOSS_BLOCK) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += partials[(size_t)b * 4 + k];
    }
    __shared__ double tot[4];
    block_reduce_store<4>(acc, tot);
    if (threadIdx.x == 0) {
        const double n = (double)B;
        out4[0] = (float)(-tot[0] / n);
        out4[1] = (float)(0.5 * tot[1] / n);
        out4[2] = (float)(tot[2] / n);
        out4[3] = (float)(tot[3] / n);
    }
}

}  // namespace

extern "C" {

size_t ppo_loss_workspace_bytes(long B)
{
    (void)B;
    return (size_t)LOSS_MAX_BLOCKS * 5 * sizeof(double);
}

int ppo_adv_stats(const float *adv, long B, double *stats3, void *workspace, size_t workspace_bytes,
                  void *stream)
{
    PPO_REQUIRE(adv && stats3 && B >= 1);
    PPO_REQUIRE(workspace && workspace_bytes >= ppo_loss_workspace_bytes(B) && ppo_aligned(workspace, 8));
    PPO_REQUIRE(ppo_aligned(adv, 4) && ppo_aligned(stats3, 8));
    hipStream_t s = (hipStream_t)stream;
    double *partials = reinterpret_cast<double *>(workspace);
    const int grid = loss_grid(B);
    const bool vec = ppo_aligned(adv, 16);
    hipLaunchKernelGGL(k_adv_stats_partial, dim3(grid), dim3(LOSS_BLOCK), 0, s, adv, B, partials, vec);
    int rc = ppo_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(k_adv_stats_final, dim3(1), dim3(LOSS_BLOCK), 0, s, adv, (const long *)nullptr, B, partials, grid, stats3);
    return ppo_launch_status();
}

int ppo_adv_stats_gather(const float *adv, const long *idx, long B, double *stats3, void *workspace,
                         size_t workspace_bytes, void *stream)
{
    PPO_REQUIRE(adv && idx && stats3 && B >= 1);
    PPO_REQUIRE(workspace && workspace_bytes >= ppo_loss_workspace_bytes(B) && ppo_aligned(workspace, 8));
    PPO_REQUIRE(ppo_aligned(adv, 4) && ppo_aligned(idx, 8) && ppo_aligned(stats3, 8));
    hipStream_t s = (hipStream_t)stream;
    double *partials = reinterpret_cast<double *>(workspace);
    const int grid = loss_grid(B);
    hipLaunchKernelGGL(k_adv_stats_partial_idx, dim3(grid), dim3(LOSS_BLOCK), 0, s, adv, idx, B, partials);
    int rc = ppo_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(k_adv_stats_final, dim3(1), dim3(LOSS_BLOCK), 0, s, adv, idx, B, partials, grid, stats3);
    return ppo_launch_status();
}

int ppo_loss_fwd_bwd(const float *x, const float *old_lp, const float *new_v, const float *old_v,
                     const float *adv, const double *stats3, double clip, int norm_adv, int clip_va,
                     int ratio_input, long B, float *out4, float *g_x, float *g_v, void *workspace,
                     size_t workspace_bytes, void *stream)
{
    PPO_REQUIRE(x && new_v && old_v && adv && out4 && B >= 1);
    PPO_REQUIRE(ratio_input || old_lp);
    PPO_REQUIRE(!norm_adv || stats3);
    PPO_REQUIRE(workspace && workspace_bytes >= ppo_loss_workspace_bytes(B) && ppo_aligned(workspace, 8));
    hipStream_t s = (hipStream_t)stream;
    LossArgs a;
    a.x = x; a.old_lp = old_lp; a.new_v = new_v; a.old_v = old_v; a.adv = adv; a.stats3 = stats3;
    a.g_x = g_x; a.g_v = g_v; a.partials = reinterpret_cast<double *>(workspace);
    a.lo = (float)(1.0 - clip); a.hi = (float)(1.0 + clip); a.epsf = (float)clip;
    a.B = B; a.norm_adv = norm_adv; a.clip_va = clip_va; a.ratio_input = ratio_input;
    a.vec = ppo_aligned(x, 16) && ppo_aligned(new_v, 16) && ppo_aligned(old_v, 16) && ppo_aligned(adv, 16) &&
            (ratio_input || ppo_aligned(old_lp, 16)) && (!g_x || ppo_aligned(g_x, 16)) &&
            (!g_v || ppo_aligned(g_v, 16));
    const int grid = loss_grid(B);
    hipLaunchKernelGGL(k_loss_partial, dim3(grid), dim3(LOSS_BLOCK), 0, s, a);
    int rc = ppo_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(LOSS_BLOCK), 0, s, a.partials, grid, B, out4);
    return ppo_launch_status();
}

int ppo_loss_logits_fwd_bwd(const float *logits, int num_actions, const int *actions, const float *old_lp,
                            const float *new_v, const float *old_v, const float *adv, const double *stats3,
                            double clip, int norm_adv, int clip_va, long B, float *out5, float *g_logits_pl,
                            float *g_logits_ent, float *g_v, void *workspace, size_t workspace_bytes, void *stream)
{
    PPO_REQUIRE(logits && actions && old_lp && new_v && old_v && adv && out5 && B >= 1);
    PPO_REQUIRE(num_actions >= 2 && num_actions <= 8);
    PPO_REQUIRE(!norm_adv || stats3);
    PPO_REQUIRE(workspace && workspace_bytes >= ppo_loss_workspace_bytes(B) && ppo_aligned(workspace, 8));
    hipStream_t s = (hipStream_t)stream;
    LossLogitsArgs a;
    a.logits = logits; a.actions = actions; a.old_lp = old_lp; a.new_v = new_v; a.old_v = old_v; a.adv = adv;
    a.stats3 = stats3; a.g_logits_pl = g_logits_pl; a.g_logits_ent = g_logits_ent; a.g_v = g_v;
    a.partials = reinterpret_cast<double *>(workspace);
    a.lo = (float)(1.0 - clip); a.hi = (float)(1.0 + clip); a.epsf = (float)clip;
    a.B = B; a.norm_adv = norm_adv; a.clip_va = clip_va;
    long blocks = (B + LOSS_BLOCK - 1) / LOSS_BLOCK;
    const int grid = (int)(blocks > LOSS_MAX_BLOCKS ? LOSS_MAX_BLOCKS : blocks);
    switch (num_actions) {
#define PPO_CASE(A_) case A_: hipLaunchKernelGGL(k_loss_logits_partial<A_>, dim3(grid), dim3(LOSS_BLOCK), 0, s, a); break;
        PPO_CASE(2) PPO_CASE(3) PPO_CASE(4) PPO_CASE(5) PPO_CASE(6) PPO_CASE(7) PPO_CASE(8)
#undef PPO_CASE
    }
    int rc = ppo_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(k_loss_logits_final, dim3(1), dim3(LOSS_BLOCK), 0, s, a.partials, grid, B, out5);
    return ppo_launch_status();
}

}  // extern "C"
