// Episode statistics on the device rings -- gfx950.
//
// Replaces the tail of SyncVecEnv.rollout (reference src/ppo.py:404-417): concatenating every
// env's RecordEpisodeStatistics deques (last <= 100 finished episodes), np.max and
// np.percentile([5, 50, 95]) of the returns, np.percentile(50) of the lengths.
//
// Only nine order statistics of up to N * 100 values are needed (two neighbours per percentile for
// np.percentile's "linear" interpolation, plus the maximum), so nothing is sorted: a three-level
// radix SELECT (12 + 10 + 10 key bits) walks the rings three times, each time histogramming only
// the digit that still separates the wanted ranks, and a one-block kernel after each pass narrows
// every rank to its bucket.  Exact (the selected keys are the values themselves).  Traffic: 3 reads
// of the valid ring slots (<= 52 MB each at N = 65536); an earlier version materialised +inf-padded
// key arrays and radix-sorted them (rocPRIM via torch.sort): ~460 us of kernel time per rollout,
// against ~60 us here.
//   k_sel_hist<L>  : rings -> LDS histograms -> global histogram(s) of level L
//   k_sel_scan<L>  : histogram(s) -> bucket and residual rank of every target; the last level
//                    writes {running max, p5, p50, p95, median length} (f64)
#include "common.h"

namespace {

constexpr int SEL_T = 9;          // targets: returns lo/hi of p5, p50, p95; returns max; lengths lo/hi of p50
constexpr int SEL_T_RET = 7;      // targets 0..6 select from the returns, 7..8 from the lengths
constexpr int SEL_B0 = 4096;      // level-0 buckets (top 12 key bits)
constexpr int SEL_B1 = 1024;      // level-1 / level-2 buckets (10 bits each)
constexpr int SEL_BLOCK = 256;       // scan kernels
constexpr int SEL_HBLOCK = 1024;     // histogram kernels: 16 waves share one LDS histogram, 2 blocks fill a CU

struct SelState {                 // lives at the head of the workspace
    unsigned long long n;         // number of valid ring entries (same for returns and lengths)
    unsigned long long rank[SEL_T];   // residual rank of target t inside its current bucket
    uint32_t prefix[SEL_T];       // key bits fixed so far (12, then 22, then 32)
    int slot[SEL_T];              // histogram this target reads at the next level (targets that share a prefix share one)
    int leader[SEL_T];            // 1 if slot[t] == t
};

struct SelWs {
    SelState *st;
    uint32_t *hist0;              // [2][SEL_B0]
    uint32_t *hist1;              // [SEL_T][SEL_B1]
    uint32_t *hist2;              // [SEL_T][SEL_B1]
};

constexpr size_t SEL_STATE_BYTES = 512;
static_assert(sizeof(SelState) <= SEL_STATE_BYTES, "state block too small");
constexpr size_t SEL_WS_BYTES = SEL_STATE_BYTES + (2 * SEL_B0 + 2 * SEL_T * SEL_B1) * sizeof(uint32_t);

inline SelWs sel_ws(void *workspace)
{
    SelWs w;
    char *p = static_cast<char *>(workspace);
    w.st = reinterpret_cast<SelState *>(p);
    w.hist0 = reinterpret_cast<uint32_t *>(p + SEL_STATE_BYTES);
    w.hist1 = w.hist0 + 2 * SEL_B0;
    w.hist2 = w.hist1 + SEL_T * SEL_B1;
    return w;
}

// f32 -> u32 whose unsigned order is the float order
__device__ __forceinline__ uint32_t f2key(float x)
{
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

template <int LEVEL>
__global__ void __launch_bounds__(SEL_HBLOCK) k_sel_hist(const float *ret_ring, const int *len_ring, const int *ring_cnt,
                                                        int ring, int N, SelWs w)
{
    constexpr int NH = LEVEL == 0 ? 2 : SEL_T, NB = LEVEL == 0 ? SEL_B0 : SEL_B1;
    constexpr int SHIFT = LEVEL == 0 ? 20 : (LEVEL == 1 ? 10 : 0);     // digit position
    constexpr int PSHIFT = LEVEL == 1 ? 20 : 10;                       // bits below the fixed prefix
    __shared__ uint32_t h[NH * NB];
    for (int k = threadIdx.x; k < NH * NB; k += SEL_HBLOCK) h[k] = 0u;
    uint32_t prefix[SEL_T];
    bool lead[SEL_T];
    if (LEVEL > 0) {
        if (w.st->n == 0) return;                                      // uniform: nothing finished yet
#pragma unroll
        for (int t = 0; t < SEL_T; ++t) { prefix[t] = w.st->prefix[t]; lead[t] = w.st->leader[t] != 0; }
    }
    __syncthreads();
    // one ring (env) per wave at a time, lanes across its slots: coalesced, and no per-element
    // division by the run-time ring length
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (SEL_HBLOCK / 64) + (threadIdx.x >> 6), nwaves = gridDim.x * (SEL_HBLOCK / 64);
    for (int n = wave; n < N; n += nwaves) {
        const int c0 = ring_cnt[n], c = c0 < ring ? c0 : ring;
        const size_t row = (size_t)n * ring;
        for (int k = lane; k < c; k += 64) {
            const uint32_t kr = f2key(ret_ring[row + k]), kl = f2key((float)len_ring[row + k]);
            if (LEVEL == 0) {
                atomicAdd(&h[kr >> SHIFT], 1u);
                atomicAdd(&h[SEL_B0 + (kl >> SHIFT)], 1u);
            } else {
#pragma unroll
                for (int t = 0; t < SEL_T; ++t) {
                    const uint32_t key = t < SEL_T_RET ? kr : kl;
                    if (lead[t] && (key >> PSHIFT) == prefix[t]) atomicAdd(&h[t * NB + ((key >> SHIFT) & (NB - 1))], 1u);
                }
            }
        }
    }
    __syncthreads();
    uint32_t *g = LEVEL == 0 ? w.hist0 : (LEVEL == 1 ? w.hist1 : w.hist2);
    for (int k = threadIdx.x; k < NH * NB; k += SEL_HBLOCK) {
        const uint32_t v = h[k];
        if (v) atomicAdd(&g[k], v);       // integer adds: order-independent, deterministic
    }
}

// Inclusive scan of nb (<= 4096, multiple of SEL_BLOCK) bins into cum[] (LDS); returns the total.
__device__ unsigned long long block_scan_bins(const uint32_t *bins, int nb, unsigned long long *cum,
                                              unsigned long long *part)
{
    const int per = nb / SEL_BLOCK, base = threadIdx.x * per;
    unsigned long long s = 0;
    for (int k = 0; k < per; ++k) s += bins[base + k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < SEL_BLOCK; off <<= 1) {     // Hillis-Steele over the 256 partial sums
        const unsigned long long add = threadIdx.x >= off ? part[threadIdx.x - off] : 0ull;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned long long run = threadIdx.x ? part[threadIdx.x - 1] : 0ull;
    for (int k = 0; k < per; ++k) {
        run += bins[base + k];
        cum[base + k] = run;
    }
    __syncthreads();
    return part[SEL_BLOCK - 1];
}

// smallest b with cum[b] > rank
__device__ int find_bucket(const unsigned long long *cum, int nb, unsigned long long rank)
{
    int lo = 0, hi = nb - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] > rank) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__device__ void rank_pair(unsigned long long n, double q, unsigned long long &lo, unsigned long long &hi)
{
    const double pos = q * (double)(n - 1);                 // np.percentile, method "linear"
    lo = (unsigned long long)floor(pos);
    hi = lo + 1 < n ? lo + 1 : n - 1;
}

__device__ double lerp_pair(double a, double b, unsigned long long n, double q)
{
    const double pos = q * (double)(n - 1);
    return a + (b - a) * (pos - floor(pos));
}

// stats5 = {running max, p5, p50, p95, median length}; entries keep their previous value (NaN at
// start) while no episode has finished, like the reference's attributes (src/ppo.py:250-254, :408-417).
template <int LEVEL> __global__ void __launch_bounds__(SEL_BLOCK) k_sel_scan(SelWs w, double *stats5)
{
    __shared__ unsigned long long cum[SEL_B0];
    __shared__ unsigned long long part[SEL_BLOCK];
    __shared__ unsigned long long s_rank[SEL_T];            // the block works on an LDS copy of the state
    __shared__ uint32_t s_prefix[SEL_T];
    __shared__ int s_slot[SEL_T], s_leader[SEL_T];
    SelState *st = w.st;
    const int tid = threadIdx.x;
    unsigned long long n;
    if (LEVEL == 0) {
        n = block_scan_bins(w.hist0, SEL_B0, cum, part);
        if (n == 0) {                                        // uniform: nothing finished yet
            if (tid == 0) st->n = 0;
            return;
        }
        if (tid == 0) {
            rank_pair(n, 0.05, s_rank[0], s_rank[1]);
            rank_pair(n, 0.50, s_rank[2], s_rank[3]);
            rank_pair(n, 0.95, s_rank[4], s_rank[5]);
            s_rank[6] = n - 1;
            rank_pair(n, 0.50, s_rank[7], s_rank[8]);
        }
        __syncthreads();
        if (tid < SEL_T_RET) {
            const int b = find_bucket(cum, SEL_B0, s_rank[tid]);
            s_rank[tid] -= b ? cum[b - 1] : 0ull;
            s_prefix[tid] = (uint32_t)b;
        }
        __syncthreads();
        block_scan_bins(w.hist0 + SEL_B0, SEL_B0, cum, part);
        if (tid >= SEL_T_RET && tid < SEL_T) {
            const int b = find_bucket(cum, SEL_B0, s_rank[tid]);
            s_rank[tid] -= b ? cum[b - 1] : 0ull;
            s_prefix[tid] = (uint32_t)b;
        }
    } else {
        n = st->n;
        if (n == 0) return;
        if (tid < SEL_T) {
            s_rank[tid] = st->rank[tid]; s_prefix[tid] = st->prefix[tid];
            s_slot[tid] = st->slot[tid]; s_leader[tid] = st->leader[tid];
        }
        __syncthreads();
        const uint32_t *hist = LEVEL == 1 ? w.hist1 : w.hist2;
        for (int t = 0; t < SEL_T; ++t) {                    // one 1024-bin scan per distinct prefix
            if (!s_leader[t]) continue;                      // uniform across the block
            block_scan_bins(hist + t * SEL_B1, SEL_B1, cum, part);
            if (tid < SEL_T && s_slot[tid] == t) {
                const int b = find_bucket(cum, SEL_B1, s_rank[tid]);
                s_rank[tid] -= b ? cum[b - 1] : 0ull;
                s_prefix[tid] = (s_prefix[tid] << 10) | (uint32_t)b;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid != 0) return;
    if (LEVEL < 2) {                                         // targets with equal prefixes share a histogram
        st->n = n;
        for (int t = 0; t < SEL_T; ++t) {
            int s = t;
            for (int u = (t < SEL_T_RET ? 0 : SEL_T_RET); u < t; ++u)
                if (s_prefix[u] == s_prefix[t]) { s = u; break; }
            st->slot[t] = s;
            st->leader[t] = s == t;
            st->rank[t] = s_rank[t];
            st->prefix[t] = s_prefix[t];
        }
    } else {
        double v[SEL_T];
        for (int t = 0; t < SEL_T; ++t) v[t] = (double)key2f(s_prefix[t]);
        if (!(stats5[0] >= v[6])) stats5[0] = v[6];          // also replaces the initial NaN
        stats5[1] = lerp_pair(v[0], v[1], n, 0.05);
        stats5[2] = lerp_pair(v[2], v[3], n, 0.50);
        stats5[3] = lerp_pair(v[4], v[5], n, 0.95);
        stats5[4] = lerp_pair(v[7], v[8], n, 0.50);
    }
}

template <int LEVEL>
int sel_level(const float *ret_ring, const int *len_ring, const int *ring_cnt, int ring, int N, const SelWs &w,
              double *stats5, unsigned blocks, hipStream_t s)
{
    hipLaunchKernelGGL(k_sel_hist<LEVEL>, dim3(blocks), dim3(SEL_HBLOCK), 0, s, ret_ring, len_ring, ring_cnt, ring, N, w);
    int rc = ppo_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(k_sel_scan<LEVEL>, dim3(1), dim3(SEL_BLOCK), 0, s, w, stats5);
    return ppo_launch_status();
}

}  // namespace

extern "C" {

size_t ppo_episode_stats_workspace_bytes(void) { return SEL_WS_BYTES; }

int ppo_episode_stats(const float *ret_ring, const int *len_ring, const int *ring_cnt, int ring, int N,
                      double *stats5, void *workspace, size_t workspace_bytes, void *stream)
{
    PPO_REQUIRE(N >= 0 && ring > 0 && (unsigned long long)N * (unsigned long long)ring < 0xFFFFFFFFull);
    PPO_REQUIRE(stats5 && ppo_aligned(stats5, 8));
    PPO_REQUIRE(workspace && workspace_bytes >= SEL_WS_BYTES && ppo_aligned(workspace, 8));
    if (N == 0) return 0;
    PPO_REQUIRE(ret_ring && len_ring && ring_cnt);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(workspace, 0, SEL_WS_BYTES, s);
    if (e != hipSuccess) return (int)e;
    const SelWs w = sel_ws(workspace);
    size_t blocks = ((size_t)N + (SEL_HBLOCK / 64) * 4 - 1) / ((SEL_HBLOCK / 64) * 4);   // >= 4 rings per wave
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    int rc = sel_level<0>(ret_ring, len_ring, ring_cnt, ring, N, w, stats5, (unsigned)blocks, s);
    if (rc) return rc;
    rc = sel_level<1>(ret_ring, len_ring, ring_cnt, ring, N, w, stats5, (unsigned)blocks, s);
    if (rc) return rc;
    return sel_level<2>(ret_ring, len_ring, ring_cnt, ring, N, w, stats5, (unsigned)blocks, s);
}

}  // extern "C"
