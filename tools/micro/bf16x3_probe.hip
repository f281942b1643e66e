// Probe for the three-piece bf16 operand split of PPO_NET_BF16X3 (csrc/mfma_split.h): is
// x == b0 + b1 + b2 EXACTLY for f32 x, for several ways of forming the residual r = x - b0 on gfx950?
//   A  __builtin_amdgcn_fdot2_f32_bf16 with a constant selector pair (hipcc folds it into an inline operand)
//   B  the builtin, selector pair in an SGPR the compiler cannot see through
//   C  the builtin, selector pair in a VGPR the compiler cannot see through
//      (round 3 also tried v_dot2c_f32_bf16 as inline asm: wrong results, a hazard the assembler does not cover)
//   D  shift / mask the piece back to f32 and subtract (plain C)
//   E  the shipped split3_bf16 (mfma_split.h)
// build: hipcc --offload-arch=gfx950 -O3 -I ppo_amd/csrc -o bf16x3_probe tools/micro/bf16x3_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "mfma_split.h"
using namespace ppo_net;

template <int V> __device__ float resid(float x, uint32_t p, bool hi)
{
    if (V == 0) {
        const bf2 lo_sel = {(__bf16)-1.0f, (__bf16)0.0f}, hi_sel = {(__bf16)0.0f, (__bf16)-1.0f};
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p), hi ? hi_sel : lo_sel, x, false);
    } else if (V == 1) {
        uint32_t sel;
        if (hi) asm("s_mov_b32 %0, 0xbf800000" : "=s"(sel)); else asm("s_mov_b32 %0, 0xbf80" : "=s"(sel));
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p), __builtin_bit_cast(bf2, sel), x, false);
    } else if (V == 2) {
        uint32_t sel;
        if (hi) asm("v_mov_b32 %0, 0xbf800000" : "=v"(sel)); else asm("v_mov_b32 %0, 0xbf80" : "=v"(sel));
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p), __builtin_bit_cast(bf2, sel), x, false);
    } else {
        const float b = __uint_as_float(hi ? (p & 0xffff0000u) : (p << 16));
        return x - b;
    }
}

template <int V> __global__ void k_split(const float *x, uint32_t *p, int n2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const float a0 = x[2 * i], a1 = x[2 * i + 1];
    uint32_t p0, p1, p2;
    if (V == 4) {
        Bf16Select sel;
        sel.init();
        split3_bf16(a0, a1, sel, p0, p1, p2);
    } else {
        p0 = pack_bf16(a0, a1);
        const float r0 = resid<V>(a0, p0, false), r1 = resid<V>(a1, p0, true);
        p1 = pack_bf16(r0, r1);
        const float s0 = resid<V>(r0, p1, false), s1 = resid<V>(r1, p1, true);
        p2 = pack_bf16(s0, s1);
    }
    if (V == 4 && i == 0) {                  // a compile-time constant input pair, as the kernel's bias slot (1.0, 0.0)
        Bf16Select sel;
        sel.init();
        uint32_t c0, c1, c2;
        split3_bf16(1.0f, 0.0f, sel, c0, c1, c2);
        p[3 * n2] = c0; p[3 * n2 + 1] = c1; p[3 * n2 + 2] = c2;
    }
    p[3 * i] = p0; p[3 * i + 1] = p1; p[3 * i + 2] = p2;
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
    (void)hipMalloc(&dx, 2 * n2 * 4); (void)hipMalloc(&dp, 3 * n2 * 4 + 12);
    (void)hipMemcpy(dx, h.data(), 2 * n2 * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> p(3 * n2 + 3);
    int rc = 0;
    for (int v = 0; v < 5; ++v) {
        (void)hipMemset(dp, 0, 3 * n2 * 4);
        if (v == 0) hipLaunchKernelGGL(k_split<0>, dim3(n2 / 256), dim3(256), 0, 0, dx, dp, n2);
        if (v == 1) hipLaunchKernelGGL(k_split<1>, dim3(n2 / 256), dim3(256), 0, 0, dx, dp, n2);
        if (v == 2) hipLaunchKernelGGL(k_split<2>, dim3(n2 / 256), dim3(256), 0, 0, dx, dp, n2);
        if (v == 3) hipLaunchKernelGGL(k_split<3>, dim3(n2 / 256), dim3(256), 0, 0, dx, dp, n2);
        if (v == 4) hipLaunchKernelGGL(k_split<4>, dim3(n2 / 256), dim3(256), 0, 0, dx, dp, n2);
        (void)hipMemcpy(p.data(), dp, 3 * n2 * 4 + 12, hipMemcpyDeviceToHost);
        long bad[2] = {0, 0}; double worst = 0; int shown = 0;
        for (int i = 0; i < n2; ++i)
            for (int e = 0; e < 2; ++e) {
                const int sh = 16 * e;
                const double b0 = bf((p[3 * i] >> sh) & 0xffff), b1 = bf((p[3 * i + 1] >> sh) & 0xffff), b2 = bf((p[3 * i + 2] >> sh) & 0xffff);
                const double x = h[2 * i + e];
                if (b0 + b1 + b2 != x) {
                    ++bad[e]; worst = fmax(worst, fabs(b0 + b1 + b2 - x) / fmax(fabs(x), 1e-30));
                    if (shown++ < 3) printf("    variant %c half %d: x %.9g  pieces %.9g %.9g %.9g\n", 'A' + v, e, x, b0, b1, b2);
                }
            }
        printf("variant %c: %ld (lower half) + %ld (upper half) of %d values not reconstructed exactly (worst relative %.3g)\n",
               'A' + v, bad[0], bad[1], 2 * n2, worst);
        if (v == 4 && (bad[0] || bad[1])) rc = 1;
        if (v == 4) {
            const bool ok = p[3 * n2] == 0x00003f80u && p[3 * n2 + 1] == 0 && p[3 * n2 + 2] == 0;
            printf("variant E on the constant pair (1.0, 0.0): pieces %08x %08x %08x %s\n", p[3 * n2], p[3 * n2 + 1], p[3 * n2 + 2], ok ? "ok" : "WRONG");
            if (!ok) rc = 1;
        }
    }
    return rc;
}
