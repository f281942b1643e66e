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
//   k_sel_scan<L>  : histogram(s) -> bucket and residual rank of every target (one wave per target,
//                    shuffle scans); the last level writes {running max, p5, p50, p95, median length}
#include "common.h"

namespace {

constexpr int SEL_T = 9;          // targets: returns lo/hi of p5, p50, p95; returns max; lengths lo/hi of p50
constexpr int SEL_T_RET = 7;      // targets 0..6 select from the returns, 7..8 from the lengths
constexpr int SEL_B0 = 4096;      // level-0 buckets (top 12 key bits)
constexpr int SEL_B1 = 1024;      // level-1 / level-2 buckets (10 bits each)
constexpr int SEL_HBLOCK = 1024;     // histogram kernels: 16 waves share one LDS histogram, 2 blocks fill a CU
constexpr int SEL_U = 4;             // ring slots a lane has in flight

struct SelState {                 // lives at the head of the workspace
    unsigned long long n;         // number of valid ring entries (same for returns and lengths)
    unsigned long long rank[SEL_T];   // residual rank of target t inside its current bucket
    uint32_t prefix[SEL_T];       // key bits fixed so far (12, then 22, then 32)
    int slot[SEL_T];              // histogram this target reads at the next level (targets that share a prefix share one)
    int leader[SEL_T];            // 1 if slot[t] == t
    int nlead_ret, nlead;         // the leaders again as two compact lists (returns from entry 0, lengths from SEL_T_RET):
    uint32_t lprefix[SEL_T];      //   prefix to match,
    int lslot[SEL_T];             //   histogram to count into
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
    uint32_t lprefix[SEL_T];
    int lslot[SEL_T], nlead_ret = 0, nlead = 0;
    if (LEVEL > 0) {
        if (w.st->n == 0) return;                                      // uniform: nothing finished yet
        nlead_ret = __builtin_amdgcn_readfirstlane(w.st->nlead_ret);
        nlead = __builtin_amdgcn_readfirstlane(w.st->nlead);
#pragma unroll
        for (int q = 0; q < SEL_T; ++q) {
            lprefix[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)w.st->lprefix[q]);
            lslot[q] = __builtin_amdgcn_readfirstlane(w.st->lslot[q]);
        }
    }
    __syncthreads();
    // Flat walk over the N * ring slots: a wave takes 64 SEL_U consecutive slots at a time, lane l
    // the slots l, l + 64, ...; the (ring, index in ring) pair of the first one costs one division,
    // the others follow by adding 64 = q64 ring + r64.  A slot is valid if its index in the ring is
    // below the ring's count.  All 3 SEL_U loads are unconditional (indices clamped into the arrays)
    // and in flight together; the passes are bound by instruction issue (about 6.5 M slots at
    // N = 65536), so the per-slot work is kept short: no division, and at levels 1 and 2 only
    // the distinct prefixes still in play are compared (typically 5 of the 9 targets).
    const uint32_t total = (uint32_t)N * (uint32_t)ring, uring = (uint32_t)ring;
    const uint32_t q64 = 64u / uring, r64 = 64u - q64 * uring;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = blockIdx.x * (SEL_HBLOCK / 64) + (threadIdx.x >> 6), nwv = gridDim.x * (SEL_HBLOCK / 64);
    for (uint32_t b = wv * (64u * SEL_U); b < total; b += nwv * (64u * SEL_U)) {
        const uint32_t i0 = b + lane;
        uint32_t n = i0 / uring, k = i0 - n * uring;
        uint32_t idx[SEL_U], kk[SEL_U];
        int cnt[SEL_U];
#pragma unroll
        for (int u = 0; u < SEL_U; ++u) {
            const uint32_t i = i0 + 64u * u;
            const bool in = i < total;
            idx[u] = in ? i : total - 1;
            kk[u] = in ? k : 0xFFFFFFFFu;
            cnt[u] = ring_cnt[in ? n : 0u];
            k += r64; n += q64;
            if (k >= uring) { k -= uring; ++n; }
        }
        float r[SEL_U];
        int l[SEL_U];
#pragma unroll
        for (int u = 0; u < SEL_U; ++u) {
            r[u] = ret_ring[idx[u]];
            l[u] = len_ring[idx[u]];
        }
#pragma unroll
        for (int u = 0; u < SEL_U; ++u) {
            const bool act = kk[u] < (uint32_t)(cnt[u] < 0 ? 0 : cnt[u]);
            const uint32_t kr = f2key(r[u]), kl = f2key((float)l[u]);
            if (LEVEL == 0) {
                if (act) {
                    atomicAdd(&h[kr >> SHIFT], 1u);
                    atomicAdd(&h[SEL_B0 + (kl >> SHIFT)], 1u);
                }
            } else {
                const uint32_t pr = kr >> PSHIFT, pl = kl >> PSHIFT;
                const uint32_t dr = (kr >> SHIFT) & (NB - 1), dl = (kl >> SHIFT) & (NB - 1);
                int sr = -1, sl = -1;
#pragma unroll
                for (int q = 0; q < SEL_T_RET; ++q) {  // fully unrolled: the list stays in scalar registers
                    if (q < nlead_ret && pr == lprefix[q]) sr = lslot[q];
                }
#pragma unroll
                for (int q = 0; q < SEL_T - SEL_T_RET; ++q) {
                    if (q < nlead - nlead_ret && pl == lprefix[SEL_T_RET + q]) sl = lslot[SEL_T_RET + q];
                }
                // prefixes in a list are distinct, so a key matches at most one: one LDS atomic
                // instruction per key kind (their issue rate is what the pass costs beyond its loads)
                if (act && sr >= 0) atomicAdd(&h[sr * NB + dr], 1u);
                if (act && sl >= 0) atomicAdd(&h[sl * NB + dl], 1u);
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

// One wave owns one target: lane l holds PER consecutive bins, a shuffle scan gives every lane the
// count before its bins, and the lane whose bins contain the wanted rank walks them.  No LDS, no
// block barrier: the nine targets are narrowed concurrently by nine waves.
template <int PER> struct WaveBins {
    uint32_t v[PER];
    unsigned long long excl, incl, total;
    __device__ __forceinline__ void load(const uint32_t *bins)
    {
        const int lane = threadIdx.x & 63;
        const uint4 *p = reinterpret_cast<const uint4 *>(bins + lane * PER);
        unsigned long long s = 0;
#pragma unroll
        for (int k = 0; k < PER / 4; ++k) {
            const uint4 x = p[k];
            v[4 * k] = x.x; v[4 * k + 1] = x.y; v[4 * k + 2] = x.z; v[4 * k + 3] = x.w;
            s += (unsigned long long)x.x + x.y + x.z + x.w;
        }
        incl = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        excl = incl - s;
        total = __shfl(incl, 63, 64);
    }
    // bucket holding `rank` (rank < total); rank becomes the residual rank inside that bucket
    __device__ __forceinline__ int find(unsigned long long &rank) const
    {
        const int lane = threadIdx.x & 63;
        const bool here = rank >= excl && rank < incl;
        int b = 0;
        unsigned long long resid = 0;
        if (here) {
            unsigned long long run = excl;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                if (rank >= run && rank < run + v[k]) { b = lane * PER + k; resid = rank - run; }
                run += v[k];
            }
        }
        const int src = __ffsll((long long)__ballot(here)) - 1;
        rank = __shfl(resid, src, 64);
        return __shfl(b, src, 64);
    }
};

constexpr int SEL_SBLOCK = 64 * SEL_T;   // scan kernels: one wave per target

// stats5 = {running max, p5, p50, p95, median length}; entries keep their previous value (NaN at
// start) while no episode has finished, like the reference's attributes (src/ppo.py:250-254, :408-417).
template <int LEVEL> __global__ void __launch_bounds__(SEL_SBLOCK) k_sel_scan(SelWs w, double *stats5)
{
    __shared__ unsigned long long s_rank[SEL_T];
    __shared__ uint32_t s_prefix[SEL_T];
    SelState *st = w.st;
    const int t = threadIdx.x >> 6, lane = threadIdx.x & 63;   // this wave's target
    unsigned long long n, rank;
    uint32_t prefix;
    if (LEVEL == 0) {
        WaveBins<SEL_B0 / 64> wb;
        wb.load(w.hist0 + (t < SEL_T_RET ? 0 : SEL_B0));
        n = wb.total;                                            // same for returns and lengths
        if (n == 0) {                                            // uniform: nothing finished yet
            if (threadIdx.x == 0) st->n = 0;
            return;
        }
        unsigned long long lo, hi;
        rank_pair(n, t < 2 ? 0.05 : (t < 4 || t >= SEL_T_RET ? 0.50 : 0.95), lo, hi);
        const bool upper = t < 6 ? (t & 1) : (t == 8);           // targets come in (lower, upper) pairs
        rank = t == 6 ? n - 1 : (upper ? hi : lo);
        prefix = (uint32_t)wb.find(rank);
    } else {
        n = st->n;
        if (n == 0) return;                                      // nothing was counted: all histograms are still zero
        // The histograms this level does not read are cleared here for the next call (level 1: hist0,
        // consumed by level 0's narrowing; level 2: hist1), and level 2 clears its own after the reads
        // below: a call finds the workspace zeroed and leaves it zeroed, without a memset launch.
        uint32_t *stale = LEVEL == 1 ? w.hist0 : w.hist1;
        for (int k = threadIdx.x; k < (LEVEL == 1 ? 2 * SEL_B0 : SEL_T * SEL_B1); k += SEL_SBLOCK) stale[k] = 0u;
        WaveBins<SEL_B1 / 64> wb;
        wb.load((LEVEL == 1 ? w.hist1 : w.hist2) + st->slot[t] * SEL_B1);
        rank = st->rank[t];
        prefix = (st->prefix[t] << 10) | (uint32_t)wb.find(rank);
    }
    if (lane == 0) { s_rank[t] = rank; s_prefix[t] = prefix; }
    __syncthreads();
    if (LEVEL == 2) {
        for (int k = threadIdx.x; k < SEL_T * SEL_B1; k += SEL_SBLOCK) w.hist2[k] = 0u;
    }
    if (threadIdx.x != 0) return;
    if (LEVEL < 2) {                                         // targets with equal prefixes share a histogram
        st->n = n;
        for (int u = 0; u < SEL_T; ++u) {
            int sl = u;
            for (int q = (u < SEL_T_RET ? 0 : SEL_T_RET); q < u; ++q)
                if (s_prefix[q] == s_prefix[u]) { sl = q; break; }
            st->slot[u] = sl;
            st->leader[u] = sl == u;
            st->rank[u] = s_rank[u];
            st->prefix[u] = s_prefix[u];
        }
        int nr = 0, nl = 0;                                  // returns' leaders from entry 0, lengths' from SEL_T_RET
        for (int u = 0; u < SEL_T; ++u) {
            if (!st->leader[u]) continue;
            const int at = u < SEL_T_RET ? nr++ : SEL_T_RET + nl++;
            st->lprefix[at] = s_prefix[u];
            st->lslot[at] = u;
        }
        st->nlead_ret = nr;
        st->nlead = nr + nl;
    } else {
        double v[SEL_T];
        for (int u = 0; u < SEL_T; ++u) v[u] = (double)key2f(s_prefix[u]);
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
    hipLaunchKernelGGL(k_sel_scan<LEVEL>, dim3(1), dim3(SEL_SBLOCK), 0, s, w, stats5);
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
    const SelWs w = sel_ws(workspace);               // zero-filled by the caller once; every call leaves it zeroed
    size_t blocks = ((size_t)N * ring + SEL_HBLOCK * SEL_U - 1) / (SEL_HBLOCK * SEL_U);
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    int rc = sel_level<0>(ret_ring, len_ring, ring_cnt, ring, N, w, stats5, (unsigned)blocks, s);
    if (rc) return rc;
    rc = sel_level<1>(ret_ring, len_ring, ring_cnt, ring, N, w, stats5, (unsigned)blocks, s);
    if (rc) return rc;
    return sel_level<2>(ret_ring, len_ring, ring_cnt, ring, N, w, stats5, (unsigned)blocks, s);
}

}  // extern "C"
