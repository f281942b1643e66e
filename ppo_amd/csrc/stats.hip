// Episode statistics on the device rings -- gfx950.
//
// Replaces the tail of SyncVecEnv.rollout (reference src/ppo.py:404-417): concatenating every
// env's RecordEpisodeStatistics deques (last <= 100 finished episodes), np.max and
// np.percentile([5, 50, 95]) of the returns, np.percentile(50) of the lengths.
//   k_ring_keys      : rings -> two flat f32 key arrays (+inf where a slot is unused) + the count n
//   (caller sorts the two key arrays ascending; rocPRIM radix sort via torch.sort)
//   k_ring_quantiles : sorted keys + n -> {running max, p5, p50, p95, median length} (f64),
//                      np.percentile's default "linear" interpolation
// HBM-bound and tiny (52 MB of rings at N = 65536); exists to replace ~80 framework launches by 2.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) k_ring_keys(const float *ret_ring, const int *len_ring, const int *ring_cnt,
                                                   int ring, int N, float *ret_key, float *len_key,
                                                   unsigned long long *count)
{
    const size_t total = (size_t)N * ring;
    const float inf = __builtin_inff();
    unsigned long long local = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / ring), k = (int)(i % ring);
        const int c = ring_cnt[n] < ring ? ring_cnt[n] : ring;
        const bool valid = k < c;
        ret_key[i] = valid ? ret_ring[i] : inf;
        len_key[i] = valid ? (float)len_ring[i] : inf;
        local += valid ? 1ull : 0ull;
    }
    // one atomic per wave (integer add: order-independent, deterministic)
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off, PPO_WAVE);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
}

__device__ double lerp_rank(const float *sorted, unsigned long long n, double q)
{
    const double pos = q * (double)(n - 1);
    const unsigned long long lo = (unsigned long long)floor(pos);
    const unsigned long long hi = lo + 1 < n ? lo + 1 : n - 1;
    const double a = (double)sorted[lo], b = (double)sorted[hi];
    return a + (b - a) * (pos - (double)lo);
}

// stats5 = {running max, p5, p50, p95, median length}; entries keep their previous value (NaN at
// start) while no episode has finished, like the reference's attributes (src/ppo.py:250-254, :408-417).
__global__ void k_ring_quantiles(const float *ret_sorted, const float *len_sorted, const unsigned long long *count,
                                 double *stats5)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long n = *count;
    if (n == 0) return;
    const double mx = (double)ret_sorted[n - 1];
    if (!(stats5[0] >= mx)) stats5[0] = mx;              // also replaces the initial NaN
    stats5[1] = lerp_rank(ret_sorted, n, 0.05);
    stats5[2] = lerp_rank(ret_sorted, n, 0.50);
    stats5[3] = lerp_rank(ret_sorted, n, 0.95);
    stats5[4] = lerp_rank(len_sorted, n, 0.50);
}

}  // namespace

extern "C" {

int ppo_episode_keys(const float *ret_ring, const int *len_ring, const int *ring_cnt, int ring, int N,
                     float *ret_key, float *len_key, unsigned long long *count, void *stream)
{
    PPO_REQUIRE(N >= 0 && ring > 0);
    PPO_REQUIRE(count && ppo_aligned(count, 8));
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(count, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return (int)e;
    if (N == 0) return 0;
    PPO_REQUIRE(ret_ring && len_ring && ring_cnt && ret_key && len_key);
    const size_t total = (size_t)N * ring;
    size_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_ring_keys, dim3((unsigned)blocks), dim3(256), 0, s, ret_ring, len_ring, ring_cnt, ring, N,
                       ret_key, len_key, count);
    return ppo_launch_status();
}

int ppo_episode_stats(const float *ret_sorted, const float *len_sorted, const unsigned long long *count,
                      double *stats5, void *stream)
{
    PPO_REQUIRE(ret_sorted && len_sorted && count && stats5 && ppo_aligned(stats5, 8));
    hipLaunchKernelGGL(k_ring_quantiles, dim3(1), dim3(64), 0, (hipStream_t)stream, ret_sorted, len_sorted, count,
                       stats5);
    return ppo_launch_status();
}

}  // extern "C"
