// How much same-wave VALU work hides behind a v_mfma_f32_32x32x16_bf16 on gfx950, one wave per SIMD?
// Loop body: CH independent accumulator chains, one MFMA each in turn, followed by K independent VALU
// instructions of one kind; cycles per MFMA as a function of K.  (The whole-rollout kernel's question: its
// 120 bf16 MFMAs per wave and step showed only 27 % co-execution with its VALU work.)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_coexec tools/micro/mfma_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf8 = __attribute__((ext_vector_type(8))) __bf16;
using f2 = __attribute__((ext_vector_type(2))) float;

template <int KIND, int K, int CH> __global__ void __launch_bounds__(256) k(float *out, int iters, long long *clk)
{
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = threadIdx.x * 1e-9f + c;
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1e-3f * (threadIdx.x + i)); b[i] = (__bf16)(1e-3f * i); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 1.0f + i * 1e-3f + threadIdx.x * 1e-6f;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    float &x = v[(rep * CH * K + c * K + j) % 16];
                    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, 0.5" : "+v"(x));
                    if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                    if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x));
                }
            }
        }
    }
    const long long c1 = clock64();
    float s = 0;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}

template <int KIND, int K, int CH> void run(float *out, long long *clk)
{
    const int iters = 20000;
    hipLaunchKernelGGL((k<KIND, K, CH>), dim3(256), dim3(256), 0, 0, out, iters, clk);
    hipLaunchKernelGGL((k<KIND, K, CH>), dim3(256), dim3(256), 0, 0, out, iters, clk);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("kind %d (0 fma, 1 exp, 2 cvt_pk_bf16)  chains %d  VALU per MFMA %2d : %6.1f clock64 ticks per MFMA (x ~24 = core cycles at 2.4 GHz / 100 MHz)\n",
           KIND, CH, K, (double)c / (iters * 4.0 * CH));
}

int main()
{
    float *out; long long *clk;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&clk, 8);
    run<0, 0, 1>(out, clk); run<0, 0, 2>(out, clk); run<0, 0, 4>(out, clk);
    run<0, 2, 2>(out, clk); run<0, 4, 2>(out, clk); run<0, 6, 2>(out, clk); run<0, 8, 2>(out, clk); run<0, 12, 2>(out, clk); run<0, 16, 2>(out, clk);
    run<1, 2, 2>(out, clk); run<1, 4, 2>(out, clk); run<1, 8, 2>(out, clk);
    run<2, 4, 2>(out, clk); run<2, 8, 2>(out, clk);
    run<0, 8, 1>(out, clk); run<0, 8, 4>(out, clk);
    return 0;
}
