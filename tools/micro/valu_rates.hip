// Issue cost of single VALU instruction kinds on gfx950, one wave per SIMD (the whole-rollout kernel's
// situation): ns per instruction of a long run of INDEPENDENT instructions of one kind, relative to v_fma_f32.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/micro/valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 16
template <int KIND> __global__ void k(float *out, int iters, float seed, long long *clk)
{
    const long long c0 = clock64(), w0 = wall_clock64();
    float v[REP];
    double d[REP / 2];
    for (int i = 0; i < REP; ++i) v[i] = seed + i * 0.001f + threadIdx.x * 1e-6f;
    for (int i = 0; i < REP / 2; ++i) d[i] = v[i];
    typedef float f2 __attribute__((ext_vector_type(2)));
    for (int it = 0; it < iters; it += 8) {   // 8 x 16 instructions between branches: loop overhead out of the picture
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, 0.5" : "+v"(v[i]));
            if (KIND == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
            if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            if (KIND == 5) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(v[i]));
            if (KIND == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[i]));
            if (KIND == 7) asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) % REP]));
            if (KIND == 8) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(v[i]) : "v"(0x07060302));
            if (KIND == 9) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[i]));
        }
      }
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        if (KIND == 3) {
#pragma unroll
            for (int i = 0; i < REP; i += 2) {
                f2 x = {v[i], v[i + 1]};
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(x));
                v[i] = x[0]; v[i + 1] = x[1];
            }
        }
        if (KIND == 4) {
#pragma unroll
            for (int i = 0; i < REP / 2; ++i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i]));
        }
      }
    }
    float s = 0;
    for (int i = 0; i < REP; ++i) s += v[i];
    for (int i = 0; i < REP / 2; ++i) s += (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
static long long *g_clk;
static int g_threads = 256;
template <int KIND> double run(const char *name, int per_iter, float *out)
{
    const int iters = 400000;   // ~30 ms per kind: past the clock ramp
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipGetLastError(); hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(g_threads), 0, 0, out, iters, 1.0f, g_clk);
    (void)hipEventRecord(a);
    (void)hipGetLastError(); hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(g_threads), 0, 0, out, iters, 1.0f, g_clk);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    long long c[2]; (void)hipMemcpy(c, g_clk, 16, hipMemcpyDeviceToHost);
    const double ns = ms * 1e6 / ((double)iters * per_iter);
    printf("%-22s %7.3f ns per wave instruction, %d wave(s) per SIMD; clock64/wall_clock64 = %.2f (x100 MHz), %.1f clock64 ticks per instruction\n",
           name, ns, g_threads / 256, (double)c[0] / c[1], (double)c[0] / ((double)iters * per_iter));
    return ns;
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&g_clk, 16);
    for (int rep = 0; rep < 2; ++rep) {
        g_threads = rep == 0 ? 256 : 512;
        const double f = run<0>("v_fma_f32", REP, out);
        printf("  rcp/fma %.2f  exp/fma %.2f  pk_fma/fma %.2f  fma_f64/fma %.2f  cvt_pk_f16/fma %.2f\n",
               run<1>("v_rcp_f32", REP, out) / f, run<2>("v_exp_f32", REP, out) / f, run<3>("v_pk_fma_f32", REP / 2, out) / f,
               run<4>("v_fma_f64", REP / 2, out) / f, run<5>("v_cvt_pk_f16_f32", REP, out) / f);
    }
    g_threads = 256;
    const double f = run<0>("v_fma_f32", REP, out);
    printf("  (round 3, one wave per SIMD) cvt_pk_bf16/fma %.2f  dot2c_f32_bf16/fma %.2f  perm_b32/fma %.2f  and_b32/fma %.2f\n",
           run<6>("v_cvt_pk_bf16_f32", REP, out) / f, run<7>("v_dot2c_f32_bf16", REP, out) / f,
           run<8>("v_perm_b32", REP, out) / f, run<9>("v_and_b32", REP, out) / f);
    return 0;
}
