// Weight gradient of a small Linear layer over a huge batch -- gfx950.
//
// The update phase back-propagates through the dense 64-64 policy / value network on minibatches of
// up to millions of samples (reference src/ppo.py:606-625: loss.backward()).  For y = x W^T + b the
// gradients dW[M,K] = sum_b g[b,m] x[b,k] and db[m] = sum_b g[b,m] are reductions over the batch
// with a tiny output (M, K <= 64): a streaming, HBM-bound problem (read g and x once), which the
// BLAS back ends treat as a GEMM with K = 2M and run at 2-4 ms per layer at B = 2^21 (rocprof,
// profiles/r01: four such GEMMs were 70 % of an update).  Here: each workgroup streams a slice of
// the batch through LDS in 64-sample chunks (coalesced float4 loads), its four waves own the four
// 32x32 tiles of dW and accumulate them with v_mfma_f32_32x32x2_f32 (f32 products, the sample index
// is the MFMA k), the per-workgroup partial results go to a workspace and a second kernel adds them
// in index order (deterministic).  Algorithmic bytes: 4 (M + K) per sample.
#include "common.h"

namespace {

constexpr int WG_BLOCK = 256;      // 4 waves = the 2 x 2 tiles of a 64 x 64 output
constexpr int WG_S = 64;           // samples per chunk
constexpr int WG_LD = 68;          // LDS row stride (floats): 64 + 4 keeps float4 alignment, spreads banks
constexpr int WG_PART = 64 * 64 + 64;   // one partial: dW padded to 64 x 64, then db padded to 64
constexpr int WG_MAX_BLOCKS = 2048;    // 8 workgroups per CU: LDS admits 4 at a time, the rest queue

using f32x16 = __attribute__((ext_vector_type(16))) float;

// chunk of `rows` samples x `width` floats (contiguous in memory) -> LDS rows of stride WG_LD; rows
// beyond `rows` are zero-filled, columns beyond `width` are left alone (zeroed once by the caller)
__device__ __forceinline__ void stage(float *dst, const float *__restrict__ src, int rows, int width, bool vec)
{
    if (vec) {                                     // width % 4 == 0 and src 16-byte aligned
        const int w4 = width >> 2, total4 = WG_S * w4;
        for (int i = threadIdx.x; i < total4; i += WG_BLOCK) {
            const int b = i / w4, c = (i - b * w4) << 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < rows) v = *reinterpret_cast<const float4 *>(src + (size_t)b * width + c);
            *reinterpret_cast<float4 *>(dst + b * WG_LD + c) = v;
        }
    } else {
        const int total = WG_S * width;
        for (int i = threadIdx.x; i < total; i += WG_BLOCK) {
            const int b = i / width, c = i - b * width;
            dst[b * WG_LD + c] = b < rows ? src[(size_t)b * width + c] : 0.0f;
        }
    }
}

__global__ void __launch_bounds__(WG_BLOCK) k_wgrad_partial(const float *__restrict__ g, const float *__restrict__ x, long B,
                                                            int M, int K, long per_block, float *__restrict__ part,
                                                            bool vec_g, bool vec_x)
{
    __shared__ __attribute__((aligned(16))) float sG[WG_S * WG_LD];
    __shared__ __attribute__((aligned(16))) float sX[WG_S * WG_LD];
    for (int i = threadIdx.x; i < WG_S * WG_LD; i += WG_BLOCK) { sG[i] = 0.0f; sX[i] = 0.0f; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mt = wave >> 1, kt = wave & 1;                 // this wave's 32 x 32 tile of dW
    const bool tile_live = mt * 32 < M && kt * 32 < K;
    const int l31 = lane & 31, kk = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float bias = 0.0f;
    const long b_begin = (long)blockIdx.x * per_block;
    long b_end = b_begin + per_block;
    if (b_end > B) b_end = B;
    for (long b0 = b_begin; b0 < b_end; b0 += WG_S) {
        const int rows = (int)((b_end - b0) < WG_S ? (b_end - b0) : WG_S);
        __syncthreads();                                     // previous chunk fully consumed
        stage(sG, g + (size_t)b0 * M, rows, M, vec_g);
        stage(sX, x + (size_t)b0 * K, rows, K, vec_x);
        __syncthreads();
        if (tile_live) {
            const float *pa = sG + kk * WG_LD + mt * 32 + l31;   // A: lane = (output row, sample parity)
            const float *pb = sX + kk * WG_LD + kt * 32 + l31;   // B: lane = (sample parity, input column)
#pragma unroll 8
            for (int s = 0; s < WG_S; s += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[s * WG_LD], pb[s * WG_LD], acc, 0, 0, 0);
        }
        if (wave == 3 && lane < M) {                         // bias gradient: column sums of the chunk
#pragma unroll 8
            for (int s = 0; s < WG_S; ++s) bias += sG[s * WG_LD + lane];
        }
    }
    float *p = part + (size_t)blockIdx.x * WG_PART;
    if (tile_live) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            p[m * 64 + kt * 32 + l31] = acc[r];
        }
    }
    if (wave == 3) p[64 * 64 + lane] = lane < M ? bias : 0.0f;
}

// dW[m,k] = sum over workgroups (fixed order, f64 accumulate), db likewise.  One workgroup per 64
// consecutive outputs; its 16 waves each add every 16th partial (coalesced 256-byte reads), then the
// 16 slice sums are added in slice order.
constexpr int WG_RSLICES = 16;
__global__ void __launch_bounds__(64 * WG_RSLICES) k_wgrad_reduce(const float *__restrict__ part, int nblocks, int M, int K,
                                                                 float *__restrict__ dW, float *__restrict__ db)
{
    __shared__ double ssum[WG_RSLICES][64];
    const int e = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + e;
    const int nW = M * K;
    const bool live = i < nW + M;
    int src = 0;
    if (live) {
        if (i < nW) {
            const int m = i / K, k = i - m * K;
            src = m * 64 + k;
        } else {
            src = 64 * 64 + (i - nW);
        }
    }
    double s = 0.0;
    if (live) {
#pragma unroll 4
        for (int b = slice; b < nblocks; b += WG_RSLICES) s += (double)part[(size_t)b * WG_PART + src];
    }
    ssum[slice][e] = s;
    __syncthreads();
    if (slice == 0 && live) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < WG_RSLICES; ++q) t += ssum[q][e];
        if (i < nW) dW[i] = (float)t;
        else if (db) db[i - nW] = (float)t;
    }
}

inline int wgrad_blocks(long B)
{
    long blocks = (B + WG_S - 1) / WG_S;
    if (blocks > WG_MAX_BLOCKS) blocks = WG_MAX_BLOCKS;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace

extern "C" {

size_t ppo_linear_wgrad_workspace_bytes(long B) { return (size_t)wgrad_blocks(B) * WG_PART * sizeof(float); }

int ppo_linear_wgrad(const float *g, const float *x, long B, int M, int K, float *dW, float *db, void *workspace,
                     size_t workspace_bytes, void *stream)
{
    PPO_REQUIRE(g && x && dW && B >= 1);
    PPO_REQUIRE(M >= 1 && M <= 64 && K >= 1 && K <= 64);
    PPO_REQUIRE(workspace && workspace_bytes >= ppo_linear_wgrad_workspace_bytes(B) && ppo_aligned(workspace, 4));
    PPO_REQUIRE(ppo_aligned(g, 4) && ppo_aligned(x, 4));
    hipStream_t s = (hipStream_t)stream;
    const int blocks = wgrad_blocks(B);
    long per_block = (B + blocks - 1) / blocks;
    per_block = (per_block + WG_S - 1) / WG_S * WG_S;
    float *part = static_cast<float *>(workspace);
    const bool vec_g = (M % 4 == 0) && ppo_aligned(g, 16), vec_x = (K % 4 == 0) && ppo_aligned(x, 16);
    // tiles not covered by (M, K) are never written by the first kernel and never read by the second
    hipLaunchKernelGGL(k_wgrad_partial, dim3(blocks), dim3(WG_BLOCK), 0, s, g, x, B, M, K, per_block, part, vec_g, vec_x);
    int rc = ppo_launch_status();
    if (rc) return rc;
    const int outs = M * K + M;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((outs + 63) / 64), dim3(64 * WG_RSLICES), 0, s, part, blocks, M, K, dW, db);
    return ppo_launch_status();
}

}  // extern "C"
