// Gradient clipping + Adam step of one optimiser group in ONE launch -- gfx950.
//
// Replaces, per minibatch, `nn.utils.clip_grad_norm_(params, max_grad_norm); optimizer.step()` of the
// reference's update loop (src/ppo.py:608-609, :615-616, :625-626; optimiser = optim.Adam(lr, eps=1e-5),
// :732-733) for the small dense agents: 5-9 k parameters in 8 tensors.  PyTorch's foreach path is
// 17 launches of ~4.7 us each for that (norms, stack, norm, add, reciprocal, mul, clamp, multi-tensor
// scale, eight multi-tensor Adam pieces) and ~0.3 ms of host time, which made the update phase of the
// fused minibatch step (update.hip, 0.5 ms) host bound.  Here: one workgroup, two passes over the
// group's flat gradient bucket.
//
// Arithmetic = torch's, operation for operation, in f32 (no fused multiply-adds; the sum of squares in f64):
//   total_norm = sqrt(sum g^2);  coef = min(max_norm / (total_norm + 1e-6), 1);  g *= coef
//   m += (1 - beta1) (g - m)                       (Tensor.lerp_, weight < 0.5)
//   v  = beta2 v + (1 - beta2) g g
//   p -= step_size * (m / (sqrt(v) / sqrt(1 - beta2^t) + eps)),   step_size = lr / (1 - beta1^t)
// (torch/optim/adam.py _single_tensor_adam, torch/nn/utils/clip_grad.py).  The bias corrections are
// computed on the host in double like torch's Python scalars.  Deterministic: fixed summation order.
#include <cmath>

#include "common.h"

namespace {

constexpr int ADAM_MAX_TENSORS = 16;
constexpr int ADAM_BLOCK = 1024;

struct AdamArgs {
    float *p[ADAM_MAX_TENSORS], *m[ADAM_MAX_TENSORS], *v[ADAM_MAX_TENSORS];
    long end[ADAM_MAX_TENSORS];   // exclusive end of tensor k inside the flat gradient bucket
    int nt;
    float *g;                     // flat gradients, scaled in place by the clip coefficient
    long n;
    float max_norm;               // < 0: no clipping
    float beta2, w1, w2, eps, step_size, bc2_sqrt;   // w = 1 - beta, rounded from the double like torch's Python scalars
};

__global__ void __launch_bounds__(ADAM_BLOCK) k_clip_adam(AdamArgs a)
{
    __shared__ double part[ADAM_BLOCK / PPO_WAVE];
    __shared__ float s_coef;
    const int tid = threadIdx.x;
    float coef = 1.0f;
    if (a.max_norm >= 0.0f) {
        double s = 0.0;
        for (long i = tid; i < a.n; i += ADAM_BLOCK) {
            const double x = (double)a.g[i];
            s += x * x;
        }
        s = wave_sum(s);
        if ((tid & (PPO_WAVE - 1)) == 0) part[tid / PPO_WAVE] = s;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < ADAM_BLOCK / PPO_WAVE; ++w) t += part[w];
            const float total_norm = (float)sqrt(t);
            s_coef = fminf(a.max_norm / (total_norm + 1e-6f), 1.0f);
        }
        __syncthreads();
        coef = s_coef;
    }
    const float w1 = a.w1, w2 = a.w2;
    int k = 0;
    long begin = 0;
    for (long i = tid; i < a.n; i += ADAM_BLOCK) {
        while (i >= a.end[k]) { begin = a.end[k]; ++k; }      // i only grows: k never goes back
        const long j = i - begin;
        const float g = a.g[i] * coef;
        a.g[i] = g;
        float m = a.m[k][j], v = a.v[k][j];
        m = m + w1 * (g - m);
        v = a.beta2 * v + w2 * (g * g);
        a.m[k][j] = m;
        a.v[k][j] = v;
        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
        a.p[k][j] = a.p[k][j] - a.step_size * (m / denom);
    }
}


// The same step with ONE memory round trip, for groups of up to ADAM_SLOTS 1024-element chunks (the dense
// agents: 11 chunks): a chunk never straddles tensors, so its three pointers are wave-uniform; every
// thread requests the gradient, moments and parameter of its element of every chunk before anything is
// used, the block adds the squares, and the update runs from registers.  The generic kernel above walks
// memory three dependent times (norm pass, then load / store per 1024 elements): 15 us against 6.
constexpr int ADAM_SLOTS = 16;
struct AdamSlots {
    float *p[ADAM_SLOTS], *m[ADAM_SLOTS], *v[ADAM_SLOTS], *g[ADAM_SLOTS];   // already offset to the chunk
    int cnt[ADAM_SLOTS];
    int nslots;
    float max_norm, beta2, w1, w2, eps, step_size, bc2_sqrt;
};

__global__ void __launch_bounds__(ADAM_BLOCK) k_clip_adam_slots(AdamSlots a)
{
    __shared__ double part[ADAM_BLOCK / PPO_WAVE];
    __shared__ float s_coef;
    const int tid = threadIdx.x;
    float g[ADAM_SLOTS], m[ADAM_SLOTS], v[ADAM_SLOTS], p[ADAM_SLOTS];
#pragma unroll
    for (int s = 0; s < ADAM_SLOTS; ++s) {
        const bool on = s < a.nslots && tid < a.cnt[s];
        g[s] = on ? a.g[s][tid] : 0.0f;
        m[s] = on ? a.m[s][tid] : 0.0f;
        v[s] = on ? a.v[s][tid] : 0.0f;
        p[s] = on ? a.p[s][tid] : 0.0f;
    }
    float coef = 1.0f;
    if (a.max_norm >= 0.0f) {
        double sq = 0.0;
#pragma unroll
        for (int s = 0; s < ADAM_SLOTS; ++s) sq += (double)g[s] * (double)g[s];
        sq = wave_sum(sq);
        if ((tid & (PPO_WAVE - 1)) == 0) part[tid / PPO_WAVE] = sq;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < ADAM_BLOCK / PPO_WAVE; ++w) t += part[w];
            s_coef = fminf(a.max_norm / ((float)sqrt(t) + 1e-6f), 1.0f);
        }
        __syncthreads();
        coef = s_coef;
    }
#pragma unroll
    for (int s = 0; s < ADAM_SLOTS; ++s) {
        if (s < a.nslots && tid < a.cnt[s]) {
            const float gs = g[s] * coef;
            const float ms = m[s] + a.w1 * (gs - m[s]);
            const float vs = a.beta2 * v[s] + a.w2 * (gs * gs);
            const float denom = sqrtf(vs) / a.bc2_sqrt + a.eps;
            a.g[s][tid] = gs;
            a.m[s][tid] = ms;
            a.v[s][tid] = vs;
            a.p[s][tid] = p[s] - a.step_size * (ms / denom);
        }
    }
}

}  // namespace

extern "C" int ppo_clip_adam_step(int n_tensors, float *const *params, float *const *exp_avg,
                                  float *const *exp_avg_sq, const long *numel, float *grads,
                                  double max_grad_norm, double lr, double beta1, double beta2, double eps,
                                  long step, void *stream)
{
    PPO_REQUIRE(n_tensors >= 0 && n_tensors <= ADAM_MAX_TENSORS);
    if (n_tensors == 0) return 0;
    PPO_REQUIRE(params && exp_avg && exp_avg_sq && numel && grads);
    PPO_REQUIRE(step >= 1 && lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0);
    AdamArgs a{};
    long at = 0;
    for (int k = 0; k < n_tensors; ++k) {
        PPO_REQUIRE(params[k] && exp_avg[k] && exp_avg_sq[k] && numel[k] > 0);
        a.p[k] = params[k]; a.m[k] = exp_avg[k]; a.v[k] = exp_avg_sq[k];
        at += numel[k];
        a.end[k] = at;
    }
    PPO_REQUIRE(at <= (1L << 22));                 // one workgroup: meant for the dense agents' few thousand parameters
    a.nt = n_tensors; a.g = grads; a.n = at;
    a.max_norm = (float)max_grad_norm;
    a.beta2 = (float)beta2; a.w1 = (float)(1.0 - beta1); a.w2 = (float)(1.0 - beta2); a.eps = (float)eps;
    a.step_size = (float)(lr / (1.0 - std::pow(beta1, (double)step)));
    a.bc2_sqrt = (float)std::sqrt(1.0 - std::pow(beta2, (double)step));
    // chunks of 1024 elements that stay inside one tensor; few enough of them: the one-round-trip kernel
    AdamSlots sl{};
    int ns = 0;
    long begin = 0;
    for (int k = 0; k < n_tensors && ns <= ADAM_SLOTS; ++k) {
        for (long j = 0; j < numel[k]; j += ADAM_BLOCK) {
            if (ns < ADAM_SLOTS) {
                sl.p[ns] = params[k] + j; sl.m[ns] = exp_avg[k] + j; sl.v[ns] = exp_avg_sq[k] + j; sl.g[ns] = grads + begin + j;
                sl.cnt[ns] = (int)(numel[k] - j < ADAM_BLOCK ? numel[k] - j : ADAM_BLOCK);
            }
            if (++ns > ADAM_SLOTS) break;
        }
        begin += numel[k];
    }
    if (ns <= ADAM_SLOTS) {
        sl.nslots = ns;
        sl.max_norm = a.max_norm; sl.beta2 = a.beta2; sl.w1 = a.w1; sl.w2 = a.w2; sl.eps = a.eps;
        sl.step_size = a.step_size; sl.bc2_sqrt = a.bc2_sqrt;
        hipLaunchKernelGGL(k_clip_adam_slots, dim3(1), dim3(ADAM_BLOCK), 0, (hipStream_t)stream, sl);
    } else {
        hipLaunchKernelGGL(k_clip_adam, dim3(1), dim3(ADAM_BLOCK), 0, (hipStream_t)stream, a);
    }
    return ppo_launch_status();
}
