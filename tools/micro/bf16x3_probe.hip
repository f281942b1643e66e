// Probe for the three-piece bf16 operand split of PPO_NET_BF16X3 (csrc/mfma_split.h): is
// x == b0 + b1 + b2 EXACTLY for f32 x when the residuals are formed with v_dot2c_f32_bf16
// (r = x - b0 as dot2((b0, b0'), (-1, 0)) + x), and what does a 6-term 32x32x16 bf16 MFMA product
// give against float64?  build: hipcc --offload-arch=gfx950 -O3 -I ppo_amd/csrc -o bf16x3_probe tools/micro/bf16x3_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "mfma_split.h"
using namespace ppo_net;

__global__ void k_split(const float *x, uint32_t *p, int n2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    uint32_t a, b, c;
    split3_bf16(x[2 * i], x[2 * i + 1], a, b, c);
    p[3 * i] = a; p[3 * i + 1] = b; p[3 * i + 2] = c;
}

static double bf(uint32_t bits16) { uint32_t u = bits16 << 16; float f; memcpy(&f, &u, 4); return (double)f; }

int main()
{
    const int n2 = 1 << 20;
    std::vector<float> h(2 * n2);
    srand(1);
    for (int i = 0; i < 2 * n2; ++i) {
        const double u = rand() / (double)RAND_MAX * 2 - 1;
        const int kind = i % 4;
        h[i] = kind == 0 ? (float)u : kind == 1 ? (float)(u * 1e-3) : kind == 2 ? (float)(u * 30.0) : (float)tanh(3 * u);
    }
    float *dx; uint32_t *dp;
    (void)hipMalloc(&dx, 2 * n2 * 4); (void)hipMalloc(&dp, 3 * n2 * 4);
    (void)hipMemcpy(dx, h.data(), 2 * n2 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_split, dim3(n2 / 256), dim3(256), 0, 0, dx, dp, n2);
    std::vector<uint32_t> p(3 * n2);
    (void)hipMemcpy(p.data(), dp, 3 * n2 * 4, hipMemcpyDeviceToHost);
    long bad = 0; double worst = 0;
    for (int i = 0; i < n2; ++i)
        for (int e = 0; e < 2; ++e) {
            const int sh = 16 * e;
            const double s = bf((p[3 * i] >> sh) & 0xffff) + bf((p[3 * i + 1] >> sh) & 0xffff) + bf((p[3 * i + 2] >> sh) & 0xffff);
            const double x = h[2 * i + e];
            if (s != x) { ++bad; worst = fmax(worst, fabs(s - x) / fmax(fabs(x), 1e-30)); }
        }
    printf("split3_bf16: %ld of %d values not reconstructed exactly (worst relative %.3g)\n", bad, 2 * n2, worst);
    return bad != 0;
}
