// Does a transcendental (v_exp_f32 / v_rcp_f32) overlap with other VALU work of the SAME wave on gfx950, one wave per
// SIMD (the whole-rollout kernel's situation)?  Times long runs of independent instructions: only transcendentals,
// only plain / packed / f64 instructions, and the two kinds alternating 1:1 and 1:2.  If the alternating run costs the
// SUM of its parts, nothing overlaps and instruction order inside the rollout step cannot buy anything.
// build: hipcc --offload-arch=gfx950 -O3 -o trans_overlap tools/micro/trans_overlap.hip ; run: ./trans_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 16
typedef float f2 __attribute__((ext_vector_type(2)));
// KIND: 0 exp only, 1 fma only, 2 exp+fma alternating, 3 exp + 2 fma, 4 pk_fma only, 5 exp+pk_fma, 6 fma_f64 only, 7 exp+fma_f64,
//       8 rcp+exp alternating (two transcendentals)
template <int KIND> __global__ void k(float *out, int iters, float seed)
{
    float v[REP], w[REP];
    double d[REP / 2];
    for (int i = 0; i < REP; ++i) { v[i] = seed + i * 0.001f + threadIdx.x * 1e-6f; w[i] = v[i] * 0.5f; }
    for (int i = 0; i < REP / 2; ++i) d[i] = v[i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
#pragma unroll
            for (int i = 0; i < REP; ++i) {
                if (KIND == 0 || KIND == 2 || KIND == 3 || KIND == 5 || KIND == 7 || KIND == 8) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                if (KIND == 1 || KIND == 2 || KIND == 3) asm volatile("v_fma_f32 %0, %0, %0, 0.5" : "+v"(w[i]));
                if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %0, 0.5" : "+v"(w[(i + 8) % REP]));
                if (KIND == 4 || KIND == 5) { f2 x = {w[i & ~1], w[i | 1]}; asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(x)); w[i & ~1] = x[0]; w[i | 1] = x[1]; }
                if (KIND == 6 || KIND == 7) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i / 2]));
                if (KIND == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(w[i]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < REP; ++i) s += v[i] + w[i];
    for (int i = 0; i < REP / 2; ++i) s += (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> double run(const char *name, float *out, int threads)
{
    const int iters = 40000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double ns = ms * 1e6 / ((double)iters * 8 * REP);     // per inner group (one exp and / or its partner instructions)
    printf("%-34s %7.3f ns per group, %d wave(s) per SIMD\n", name, ns, threads / 256);
    return ns;
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 512 * 4);
    for (int threads = 256; threads <= 512; threads += 256) {
        const double e = run<0>("v_exp_f32", out, threads), f = run<1>("v_fma_f32", out, threads);
        const double ef = run<2>("v_exp_f32 + v_fma_f32", out, threads), eff = run<3>("v_exp_f32 + 2 v_fma_f32", out, threads);
        const double p = run<4>("v_pk_fma_f32", out, threads), ep = run<5>("v_exp_f32 + v_pk_fma_f32", out, threads);
        const double d = run<6>("v_fma_f64", out, threads), ed = run<7>("v_exp_f32 + v_fma_f64", out, threads);
        const double er = run<8>("v_exp_f32 + v_rcp_f32", out, threads);
        printf("  exp+fma: %.2f of the sum of its parts; exp+2fma: %.2f; exp+pk_fma: %.2f; exp+fma_f64: %.2f; exp+rcp: %.2f of 2 exp\n",
               ef / (e + f), eff / (e + 2 * f), ep / (e + p), ed / (e + d), er / (2 * e));
    }
    return 0;
}
