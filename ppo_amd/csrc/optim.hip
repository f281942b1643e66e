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
    hipLaunchKernelGGL(k_clip_adam, dim3(1), dim3(ADAM_BLOCK), 0, (hipStream_t)stream, a);
    return ppo_launch_status();
}
