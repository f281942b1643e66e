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
//   k_sel<L> : rings -> LDS histograms -> global histogram(s) of level L; the workgroup that flushes
//              last (ticket counter in the workspace) narrows every target to its bucket and residual
//              rank (one wave per target, shuffle scans); the last level writes {running max, p5, p50,
//              p95, median length}.  Three launches; level 2 returns at once when level 1 found the
//              low 10 bits of every wanted key clear (integer-valued returns and lengths).
//
// Incremental form (ppo_episode_stats_incremental): between two calls only the episodes that finished
// in between change the rings -- ~6 of an env's 100 slots per 128-step CartPole rollout under a random
// policy, fewer once the policy holds the pole.  A caller-owned state block keeps, across calls, the
// level-0 histograms, the level-1 histograms of the nine prefixes the wanted ranks were last found
// under, a count of window entries with low float bits set, each env's ring count at the last call and
// a shadow copy of the rings (the value an overwritten slot held).  k_sel_incr visits only the slots
// written since the last call (four lanes per env), subtracts the shadow value, adds the new one and
// narrows from the persistent histograms; the two full passes above run only when a wanted rank has
// moved to a prefix that is not tracked (level 1 rebuilds those histograms) or when values are not
// integers (level 2) -- otherwise they return at once.
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
    uint32_t ticket;              // workgroups of the running pass that have flushed their histogram (0 between passes)
    uint32_t low_bits;            // level 1: nonzero if a value that matched a wanted prefix has low float bits set
    int skip2;                    // set by level 1 when low_bits == 0: every wanted key is already known, level 2 returns at once
    // incremental form only (zero = nothing tracked yet)
    int skip1;                    // level 1 returns at once: the tracked histograms already cover every wanted rank
    int ntrack_ret, ntrack;       // prefixes whose level-1 histograms persist in hist1 (same list layout as lprefix / lslot)
    uint32_t tprefix[SEL_T];
    int tslot[SEL_T];
    long long nz;                 // window entries (returns and lengths) whose float pattern has low 10 bits set
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

// One wave owns one target: lane l sums PER consecutive bins, a shuffle scan gives every lane the
// count before its bins, and the lane whose bins contain the wanted rank walks them again.  The bins
// are the workgroup's LDS copy of the finished global histogram (re-read there, not kept in
// registers: the narrowing runs inside the histogram kernel, whose register budget -- two
// 1024-thread workgroups per CU -- it must not raise; walking them in global memory instead costs a
// DRAM latency per bin, 100+ us).  The nine targets are narrowed concurrently by nine waves.
// STRIDE: words between the bins of consecutive lanes in LDS.  With 64 bins per lane (level 0) a stride
// of 64 words puts all 64 lanes of a ds_read_b128 on the same banks; the level-0 copy is therefore
// padded to 68 words per lane (SEL_PAD0).
template <int PER, int STRIDE = PER> struct WaveBins {
    const uint32_t *bins;
    unsigned long long excl, incl, total;
    __device__ __forceinline__ void load(const uint32_t *b)
    {
        const int lane = threadIdx.x & 63;
        bins = b + lane * STRIDE;
        const uint4 *p = reinterpret_cast<const uint4 *>(bins);
        unsigned long long s = 0;
#pragma unroll 4
        for (int k = 0; k < PER / 4; ++k) {
            const uint4 x = p[k];                        // LDS: ds_read_b128
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
    // bucket holding `rank` (rank < total); rank becomes the residual rank inside that bucket.
    // The lane whose PER bins hold the rank is found from the scan; its bins are then scanned by the
    // whole wave again, PER <= 64 bins = one per lane (no serial walk).
    __device__ __forceinline__ int find(unsigned long long &rank) const
    {
        static_assert(PER <= 64, "second-level scan gives one bin to every lane");
        const int lane = threadIdx.x & 63;
        const bool here = rank >= excl && rank < incl;
        const int src = __ffsll((long long)__ballot(here)) - 1;
        const unsigned long long base = __shfl(excl, src, 64);            // entries before the owning lane's bins
        const uint32_t *owner = bins + (src - lane) * STRIDE;             // this lane's `bins` is offset by lane * STRIDE
        const unsigned long long v = lane < PER ? owner[lane] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long up = __shfl_up(inc, off, 64);
            if (lane >= off) inc += up;
        }
        const unsigned long long before = base + inc - v;
        const bool mine = v > 0 && rank >= before && rank < before + v;
        const int who = __ffsll((long long)__ballot(mine)) - 1;
        rank = rank - __shfl(before, who, 64);
        return src * PER + who;
    }
};

// The finished global histogram into the last workgroup's LDS.  The words were produced by device-scope
// atomics, so they are read with device-scope atomic loads (a plain load may hit a stale line in this
// XCD's L2); all of a thread's loads are issued (sel_fetch) before the first is consumed (sel_store) --
// one round trip, not one per word (a load / LDS-store loop cost ~0.6 us per iteration here).
// PAD0: the level-0 layout of the copy, 64-word chunks 68 words apart (see WaveBins).
constexpr int SEL_PAD0 = 68;
constexpr int SEL_H0_LDS = 2 * (SEL_B0 / 64) * SEL_PAD0;      // words of a padded copy of hist0
template <int WORDS> struct SelFetch {
    static_assert(WORDS % 2 == 0, "pairs");
    static constexpr int PAIRS = WORDS / 2, PER = (PAIRS + SEL_HBLOCK - 1) / SEL_HBLOCK;
    unsigned long long v[PER];
    __device__ __forceinline__ void fetch(uint32_t *g)
    {
        unsigned long long *g2 = reinterpret_cast<unsigned long long *>(g);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int k = threadIdx.x + i * SEL_HBLOCK;
            v[i] = k < PAIRS ? __hip_atomic_load(&g2[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
    }
    template <bool PAD0> __device__ __forceinline__ void store(uint32_t *lds) const
    {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int k = threadIdx.x + i * SEL_HBLOCK;
            const int word = PAD0 ? 2 * k + ((2 * k) >> 6) * (SEL_PAD0 - 64) : 2 * k;
            if (k < PAIRS) *reinterpret_cast<unsigned long long *>(lds + word) = v[i];
        }
    }
};

constexpr int SEL_SWAVES = SEL_T;        // narrowing: one wave per target, the first nine waves of the workgroup

// Hands the next level its work (one thread): targets with equal prefixes share a histogram; the
// distinct prefixes go into two compact lists (returns from entry 0, lengths from SEL_T_RET).
// `track`: the incremental form remembers the same lists as the prefixes whose level-1 histograms persist.
__device__ void sel_setup_next(SelState *st, const uint32_t *prefix, const unsigned long long *rank,
                               unsigned long long n, bool track)
{
    st->n = n;
    for (int u = 0; u < SEL_T; ++u) {
        int sl = u;
        for (int q = (u < SEL_T_RET ? 0 : SEL_T_RET); q < u; ++q)
            if (prefix[q] == prefix[u]) { sl = q; break; }
        st->slot[u] = sl;
        st->leader[u] = sl == u;
        st->rank[u] = rank[u];
        st->prefix[u] = prefix[u];
    }
    int nr = 0, nl = 0;
    for (int u = 0; u < SEL_T; ++u) {
        if (!st->leader[u]) continue;
        const int at = u < SEL_T_RET ? nr++ : SEL_T_RET + nl++;
        st->lprefix[at] = prefix[u];
        st->lslot[at] = u;
        if (track) { st->tprefix[at] = prefix[u]; st->tslot[at] = u; }
    }
    st->nlead_ret = nr;
    st->nlead = nr + nl;
    if (track) { st->ntrack_ret = nr; st->ntrack = nr + nl; }
}

// The nine selected keys -> stats5 = {running max, p5, p50, p95, median length} (one thread).
// `bits`: key bits known (22 when the floats' low 10 bits are zero, i.e. the keys' are 0 for values >= 0
// -- key top bit set -- or all ones for negative values, whose key is the complement; else 32).
__device__ void sel_finalize(double *stats5, const uint32_t *prefix, int bits, unsigned long long n)
{
    double v[SEL_T];
    for (int u = 0; u < SEL_T; ++u) {
        uint32_t key = prefix[u];
        if (bits == 22) key = (key << 10) | ((key >> 21) & 1u ? 0u : 1023u);
        v[u] = (double)key2f(key);
    }
    if (!(stats5[0] >= v[6])) stats5[0] = v[6];              // also replaces the initial NaN
    stats5[1] = lerp_pair(v[0], v[1], n, 0.05);
    stats5[2] = lerp_pair(v[2], v[3], n, 0.50);
    stats5[3] = lerp_pair(v[4], v[5], n, 0.95);
    stats5[4] = lerp_pair(v[7], v[8], n, 0.50);
}

// ppo_episode_stats_incremental_ints: the caller promised small integers and the window holds a value that is not one
// -- the statistics cannot be made from the direct histogram, so they say so: NaN (the running maximum too; it
// recovers with the next good call), never a silently wrong value.
__device__ void sel_poison(double *stats5)
{
    for (int k = 0; k < 5; ++k) stats5[k] = __longlong_as_double(0x7ff8000000000000ll);
}

// rank of target t among n values: (lower, upper) neighbours of p5, p50, p95 of the returns, their
// maximum, (lower, upper) of the median length
__device__ unsigned long long sel_target_rank(int t, unsigned long long n)
{
    unsigned long long lo, hi;
    rank_pair(n, t < 2 ? 0.05 : (t < 4 || t >= SEL_T_RET ? 0.50 : 0.95), lo, hi);
    const bool upper = t < 6 ? (t & 1) : (t == 8);
    return t == 6 ? n - 1 : (upper ? hi : lo);
}

// Narrowing of one level, run by the LAST workgroup of that level's histogram pass to finish (every
// thread of the workgroup calls it).  stats5 entries keep their previous value (NaN at start) while
// no episode has finished, like the reference's attributes (src/ppo.py:250-254, :408-417).
// INCR (incremental form): hist0 and hist1 persist between calls and are never cleared here, and
// whether level 2 is needed comes from the state's running count `nz` instead of this pass.
template <int LEVEL, bool INCR> __device__ void sel_narrow(const SelWs &w, const uint32_t *hist, double *stats5)
{
    __shared__ unsigned long long s_rank[SEL_T];
    __shared__ uint32_t s_prefix[SEL_T];
    __shared__ unsigned long long s_n;
    SelState *st = w.st;
    const int t = threadIdx.x >> 6, lane = threadIdx.x & 63;   // waves 0..8: this wave's target
    const bool worker = t < SEL_SWAVES;
    unsigned long long n = 0, rank = 0;
    uint32_t prefix = 0;
    if (LEVEL == 0) {
        if (worker) {
            WaveBins<SEL_B0 / 64, SEL_PAD0> wb;
            wb.load(hist + (t < SEL_T_RET ? 0 : SEL_H0_LDS / 2));
            n = wb.total;                                        // same for returns and lengths
            if (n > 0) {
                rank = sel_target_rank(t, n);
                prefix = (uint32_t)wb.find(rank);
            }
            if (threadIdx.x == 0) s_n = n;
        }
    } else {
        n = st->n;                                               // > 0: the pass returns at once otherwise
        if (worker) {
            WaveBins<SEL_B1 / 64> wb;
            wb.load(hist + st->slot[t] * SEL_B1);
            rank = st->rank[t];
            prefix = (st->prefix[t] << 10) | (uint32_t)wb.find(rank);
        }
        if (threadIdx.x == 0) s_n = n;
    }
    if (worker && lane == 0) { s_rank[t] = rank; s_prefix[t] = prefix; }
    __syncthreads();
    n = s_n;
    if (n == 0) {                                                // uniform: nothing finished yet (level 0 only)
        if (threadIdx.x == 0) { st->n = 0; st->low_bits = 0u; st->skip2 = 0; }
        return;
    }
    // Level 1 ends the selection when every value that matched a wanted 12-bit prefix has the low 10
    // bits of its float pattern clear (integer-valued returns and lengths -- CartPole, Acrobot: always):
    // the 22 bits known now determine the whole key, and level 2 returns at once.
    bool last = LEVEL == 2;
    if (LEVEL == 1) {
        last = INCR ? __hip_atomic_load(&st->nz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0
                    : __hip_atomic_load(&st->low_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
    }
    // Histograms are cleared here for the next call as soon as they have been consumed (level 1:
    // hist0, read by level 0's narrowing; the last level: hist1, and hist2 if level 2 filled it), so
    // that a call finds the workspace zeroed and leaves it zeroed without a memset launch.
    if (LEVEL == 1 && !INCR) {
        for (int k = threadIdx.x; k < 2 * SEL_B0; k += SEL_HBLOCK) w.hist0[k] = 0u;
    }
    if (LEVEL >= 1 && last && !INCR) {
        for (int k = threadIdx.x; k < SEL_T * SEL_B1; k += SEL_HBLOCK) w.hist1[k] = 0u;
    }
    if (LEVEL == 2) {
        for (int k = threadIdx.x; k < SEL_T * SEL_B1; k += SEL_HBLOCK) w.hist2[k] = 0u;
    }
    if (threadIdx.x != 0) return;
    if (LEVEL == 0) __hip_atomic_store(&st->low_bits, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (LEVEL == 1) st->skip2 = last ? 1 : 0;
    if (LEVEL == 1 && INCR) st->skip1 = 1;                   // the tracked histograms are complete again
    if (!last) sel_setup_next(st, s_prefix, s_rank, n, false);
    else sel_finalize(stats5, s_prefix, LEVEL == 1 ? 22 : 32, n);
}

// Histogram pass of one level over the rings into the workgroup's LDS histogram(s) h (zeroed by the caller):
// level 0 counts the top 12 key bits of every valid entry (returns into h[0 .. SEL_B0), lengths behind them);
// levels 1 / 2 count the next 10 bits of the entries whose higher key bits equal one of the listed prefixes
// (returns: lprefix[0 .. nlead_ret), lengths: lprefix[SEL_T_RET .. SEL_T_RET + nlead - nlead_ret)) into row lslot[q].
// `low` collects the low 10 float bits of every matched value (level 1).
template <int LEVEL>
__device__ __forceinline__ void sel_walk(const float *ret_ring, const int *len_ring, const int *ring_cnt, int ring, int N,
                                         uint32_t *h, const uint32_t (&lprefix)[SEL_T], const int (&lslot)[SEL_T],
                                         int nlead_ret, int nlead, uint32_t &low)
{
    constexpr int NB = LEVEL == 0 ? SEL_B0 : SEL_B1;
    constexpr int SHIFT = LEVEL == 0 ? 20 : (LEVEL == 1 ? 10 : 0);     // digit position
    constexpr int PSHIFT = LEVEL == 1 ? 20 : 10;                       // bits below the fixed prefix
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
                if (LEVEL == 1) {                          // low 10 bits of the FLOAT (not of the key: negative keys are complemented)
                    low |= (act && sr >= 0) ? (__float_as_uint(r[u]) & 1023u) : 0u;
                    low |= (act && sl >= 0) ? (__float_as_uint((float)l[u]) & 1023u) : 0u;
                }
            }
        }
    }
}

template <int LEVEL, bool INCR>
__global__ void __launch_bounds__(SEL_HBLOCK, 2) k_sel(const float *ret_ring, const int *len_ring, const int *ring_cnt,
                                                      int ring, int N, SelWs w, double *stats5)
{
    constexpr int NH = LEVEL == 0 ? 2 : SEL_T, NB = LEVEL == 0 ? SEL_B0 : SEL_B1;
    __shared__ uint32_t h[LEVEL == 0 ? SEL_H0_LDS : NH * NB];   // level 0: room for the padded copy the narrowing reads
    __shared__ int s_is_last;
    uint32_t lprefix[SEL_T];
    int lslot[SEL_T], nlead_ret = 0, nlead = 0;
    if (LEVEL > 0) {
        if (w.st->n == 0) return;                                      // uniform: nothing finished yet
        if (INCR && LEVEL == 1 && w.st->skip1) return;                 // uniform: every wanted prefix is tracked
        if (LEVEL == 2 && w.st->skip2) return;                         // uniform: level 1 finished the selection
        nlead_ret = __builtin_amdgcn_readfirstlane(w.st->nlead_ret);
        nlead = __builtin_amdgcn_readfirstlane(w.st->nlead);
#pragma unroll
        for (int q = 0; q < SEL_T; ++q) {
            lprefix[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)w.st->lprefix[q]);
            lslot[q] = __builtin_amdgcn_readfirstlane(w.st->lslot[q]);
        }
    }
    for (int k = threadIdx.x; k < NH * NB; k += SEL_HBLOCK) h[k] = 0u;
    __syncthreads();
    uint32_t low = 0u;                                                 // level 1: low 10 bits of the keys that matched
    sel_walk<LEVEL>(ret_ring, len_ring, ring_cnt, ring, N, h, lprefix, lslot, nlead_ret, nlead, low);
    const uint32_t lane = threadIdx.x & 63u;
    if (LEVEL == 1 && !INCR) {
        if (__any(low != 0u) && lane == 0) atomicOr(&w.st->low_bits, 1u);   // device-scope atomic, like the histogram adds
    }
    __syncthreads();
    uint32_t *g = LEVEL == 0 ? w.hist0 : (LEVEL == 1 ? w.hist1 : w.hist2);
    for (int k = threadIdx.x; k < NH * NB; k += SEL_HBLOCK) {
        const uint32_t v = h[k];
        if (v) atomicAdd(&g[k], v);       // integer adds: order-independent, deterministic
    }
    // The workgroup that flushes last narrows the ranks (one launch per level instead of a histogram
    // and a one-block kernel).  Everything exchanged between workgroups here is a device-scope atomic
    // (histogram adds, the ticket, low_bits; the last workgroup reads them back with atomic loads), so
    // no cache has to be written back or invalidated: a device-scope FENCE on this eight-XCD part is an
    // L2 write-back per workgroup, which made each pass 150 us instead of 17.  Waiting for this lane's
    // atomics to be acknowledged (vmcnt) orders them before the workgroup's ticket.
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
    __syncthreads();
    if (threadIdx.x == 0)
        s_is_last = __hip_atomic_fetch_add(&w.st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!s_is_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(&w.st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next pass
    SelFetch<NH * NB> f;                                     // the finished histogram into this workgroup's LDS
    f.fetch(g);
    __syncthreads();                                         // (level 0: every thread is done with the unpadded layout)
    f.template store<LEVEL == 0>(h);
    __syncthreads();
    sel_narrow<LEVEL, INCR>(w, h, stats5);
}

// Histogram pass alone, for a selection that is narrowed OUTSIDE this library: data-parallel ranks add up their
// histograms (one all-reduce per level) and narrow the job-wide ranks on the host (ppo_amd/ppo.py,
// SyncVecEnv.job_wide_episode_stats).  Row q < n_ret counts returns under prefixes[q], row SEL_T_RET + q < SEL_T_RET +
// n_len lengths under prefixes[SEL_T_RET + q]; word SEL_T * SEL_B1 counts matched values with low float bits set.
template <int LEVEL>
__global__ void __launch_bounds__(SEL_HBLOCK, 2) k_hist_level(const float *ret_ring, const int *len_ring, const int *ring_cnt,
                                                             int ring, int N, const uint32_t *prefixes, int n_ret, int n_len,
                                                             uint32_t *hist)
{
    __shared__ uint32_t h[SEL_T * SEL_B1];
    uint32_t lprefix[SEL_T];
    int lslot[SEL_T];
#pragma unroll
    for (int q = 0; q < SEL_T; ++q) {
        lprefix[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)prefixes[q]);
        lslot[q] = q;
    }
    for (int k = threadIdx.x; k < SEL_T * SEL_B1; k += SEL_HBLOCK) h[k] = 0u;
    __syncthreads();
    uint32_t low = 0u;
    sel_walk<LEVEL>(ret_ring, len_ring, ring_cnt, ring, N, h, lprefix, lslot, n_ret, n_ret + n_len, low);
    if (LEVEL == 1 && __any(low != 0u) && (threadIdx.x & 63) == 0) atomicAdd(&hist[SEL_T * SEL_B1], 1u);
    __syncthreads();
    for (int k = threadIdx.x; k < SEL_T * SEL_B1; k += SEL_HBLOCK) {
        const uint32_t v = h[k];
        if (v) atomicAdd(&hist[k], v);
    }
}

template <int LEVEL, bool INCR = false>
int sel_level(const float *ret_ring, const int *len_ring, const int *ring_cnt, int ring, int N, const SelWs &w,
              double *stats5, unsigned blocks, hipStream_t s)
{
    PPO_LAUNCH((k_sel<LEVEL, INCR>), dim3(blocks), dim3(SEL_HBLOCK), s, ret_ring, len_ring, ring_cnt, ring, N, w, stats5);
    return ppo_launch_status();
}

// ---- incremental form -------------------------------------------------------------------------
constexpr size_t INCR_HEAD_BYTES = (SEL_WS_BYTES + 255) / 256 * 256;

struct IncrBufs {                        // behind the histograms in the state block
    int *cnt_prev;                       // [N]        ring_cnt at the last call
    uint2 *shadow;                       // [N, ring]  the rings as they were at the last call: {return bits, length}
};

inline size_t incr_cnt_bytes(int N) { return ((size_t)N * sizeof(int) + 255) / 256 * 256; }
inline size_t incr_state_bytes(int N, int ring)
{
    return INCR_HEAD_BYTES + incr_cnt_bytes(N) + 2 * (size_t)N * (size_t)ring * sizeof(float);
}

// One value entering (sign = +1) or leaving (-1) the window: level-0 bin, the level-1 bin of a
// tracked prefix, and the count of values that need level 2.  The deltas are two's-complement adds
// into LDS words that start at zero.
__device__ __forceinline__ void incr_apply(uint32_t *h0, uint32_t *h1, uint32_t sign, float r, int l,
                                           const uint32_t (&tprefix)[SEL_T], const int (&tslot)[SEL_T],
                                           int ntrack_ret, int ntrack, int &nz)
{
    const float lf = (float)l;
    const uint32_t kr = f2key(r), kl = f2key(lf);
    atomicAdd(&h0[kr >> 20], sign);
    atomicAdd(&h0[SEL_B0 + (kl >> 20)], sign);
    const uint32_t pr = kr >> 20, pl = kl >> 20;
    int sr = -1, sl = -1;
#pragma unroll
    for (int q = 0; q < SEL_T_RET; ++q) {
        if (q < ntrack_ret && pr == tprefix[q]) sr = tslot[q];
    }
#pragma unroll
    for (int q = 0; q < SEL_T - SEL_T_RET; ++q) {
        if (q < ntrack - ntrack_ret && pl == tprefix[SEL_T_RET + q]) sl = tslot[SEL_T_RET + q];
    }
    if (sr >= 0) atomicAdd(&h1[sr * SEL_B1 + ((kr >> 10) & (SEL_B1 - 1))], sign);
    if (sl >= 0) atomicAdd(&h1[sl * SEL_B1 + ((kl >> 10) & (SEL_B1 - 1))], sign);
    nz += (int)sign * (((__float_as_uint(r) & 1023u) != 0u) + ((__float_as_uint(lf) & 1023u) != 0u));
}

constexpr int INCR_ENVS = 256;           // envs a workgroup takes per pass (their new slots are shared out over all 1024 threads)

__global__ void __launch_bounds__(SEL_HBLOCK, 1) k_sel_incr(const float *__restrict__ ret_ring, const int *__restrict__ len_ring,
                                                           const int *__restrict__ ring_cnt, int ring, int N, SelWs w,
                                                           IncrBufs ib, double *stats5)
{
    __shared__ uint32_t h0[SEL_H0_LDS];                        // deltas in the first 2 * SEL_B0 words; later the padded copy
    __shared__ uint32_t h1[SEL_T * SEL_B1];
    __shared__ int s_is_last;
    __shared__ unsigned long long s_rank[SEL_T];
    __shared__ uint32_t s_prefix[SEL_T];
    __shared__ int s_row[SEL_T];
    __shared__ unsigned long long s_n;
    __shared__ uint32_t s_first[INCR_ENVS + 1];               // work items before env e of this pass (exclusive prefix of m)
    __shared__ int s_c0[INCR_ENVS], s_c1[INCR_ENVS];
    __shared__ uint32_t s_wsum[INCR_ENVS / 64];
    SelState *st = w.st;
    uint2 *__restrict__ shadow = ib.shadow;
    uint32_t tprefix[SEL_T];
    int tslot[SEL_T];
    const int ntrack_ret = __builtin_amdgcn_readfirstlane(st->ntrack_ret);
    const int ntrack = __builtin_amdgcn_readfirstlane(st->ntrack);
#pragma unroll
    for (int q = 0; q < SEL_T; ++q) {
        tprefix[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)st->tprefix[q]);
        tslot[q] = __builtin_amdgcn_readfirstlane(st->tslot[q]);
    }
    for (int k = threadIdx.x; k < 2 * SEL_B0; k += SEL_HBLOCK) h0[k] = 0u;
    for (int k = threadIdx.x; k < SEL_T * SEL_B1; k += SEL_HBLOCK) h1[k] = 0u;

    // Slots written since the last call: entries i in [c1 - m, c1) of env n's ring, m = min(c1 - c0, ring),
    // at slot i % ring; what such a slot held before (the shadow copy) was in the old window iff the ring
    // had wrapped (c0 >= ring) or the slot was filled (slot < c0).  A pass takes 256 envs: their counts
    // are read coalesced, a prefix sum over m numbers the (env, entry) pairs, and the 1024 threads take
    // pairs round robin -- two at a time, all eight loads in flight before the first is used: the
    // accesses are scattered 4-byte reads, latency is what they cost.
    const uint32_t uring = (uint32_t)ring;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int nz = 0;
    for (int env0 = blockIdx.x * INCR_ENVS; env0 < N; env0 += gridDim.x * INCR_ENVS) {
        __syncthreads();                                       // LDS clears done / the previous pass has read s_first
        uint32_t m = 0, incl = 0;
        int c0 = 0, c1 = 0;
        if (threadIdx.x < INCR_ENVS) {
            const int n = env0 + threadIdx.x;
            if (n < N) {
                c1 = max(ring_cnt[n], 0);
                c0 = ib.cnt_prev[n];
                m = (uint32_t)max(min(c1 - c0, ring), 0);      // a count that went DOWN is not supported: treated as unchanged
                if (m > 0) ib.cnt_prev[n] = c1;
            }
            incl = m;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off, 64);
                if (lane >= off) incl += up;
            }
            if (lane == 63) s_wsum[wv] = incl;
            s_c0[threadIdx.x] = c0;
            s_c1[threadIdx.x] = c1;
        }
        __syncthreads();
        if (threadIdx.x < INCR_ENVS) {
            uint32_t before = 0;
            for (int q = 0; q < wv; ++q) before += s_wsum[q];
            s_first[threadIdx.x] = before + incl - m;
            if (threadIdx.x == INCR_ENVS - 1) s_first[INCR_ENVS] = before + incl;
        }
        __syncthreads();
        const uint32_t items = s_first[INCR_ENVS];
        for (uint32_t it0 = threadIdx.x; it0 < items; it0 += 2 * SEL_HBLOCK) {
            size_t at[2];
            bool live[2], old_valid[2];
            float r_new[2], r_old[2];
            int l_new[2], l_old[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t it = it0 + u * SEL_HBLOCK;
                live[u] = it < items;
                int e = 0;                                     // largest e with s_first[e] <= it
#pragma unroll
                for (int stepb = INCR_ENVS / 2; stepb > 0; stepb >>= 1)
                    if (s_first[e + stepb] <= it) e += stepb;
                if (!live[u]) e = 0;
                const uint32_t first = s_first[e], cnt_e = s_first[e + 1] - first;
                const int ec0 = s_c0[e], ec1 = s_c1[e];
                const uint32_t i = (uint32_t)ec1 - cnt_e + (it - first);
                const uint32_t slot = live[u] ? i % uring : 0u;
                at[u] = (size_t)(env0 + e) * uring + slot;
                old_valid[u] = live[u] && (ec0 >= ring || slot < (uint32_t)ec0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                r_new[u] = ret_ring[at[u]]; l_new[u] = len_ring[at[u]];
                const uint2 old = shadow[at[u]];
                r_old[u] = __uint_as_float(old.x); l_old[u] = (int)old.y;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (live[u]) {
                    shadow[at[u]] = make_uint2(__float_as_uint(r_new[u]), (uint32_t)l_new[u]);
                    incr_apply(h0, h1, 1u, r_new[u], l_new[u], tprefix, tslot, ntrack_ret, ntrack, nz);
                }
                if (old_valid[u]) incr_apply(h0, h1, 0xFFFFFFFFu, r_old[u], l_old[u], tprefix, tslot, ntrack_ret, ntrack, nz);
            }
        }
    }
    nz = (int)wave_sum((double)nz);
    if ((threadIdx.x & 63) == 0 && nz != 0)
        __hip_atomic_fetch_add(&st->nz, (long long)nz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * SEL_B0; k += SEL_HBLOCK) {
        const uint32_t v = h0[k];
        if (v) atomicAdd(&w.hist0[k], v);
    }
    for (int k = threadIdx.x; k < SEL_T * SEL_B1; k += SEL_HBLOCK) {
        const uint32_t v = h1[k];
        if (v) atomicAdd(&w.hist1[k], v);
    }
    // last workgroup to flush narrows (see k_sel for why atomics + vmcnt(0) and no fence)
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
    __syncthreads();
    if (threadIdx.x == 0)
        s_is_last = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!s_is_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SelFetch<2 * SEL_B0> f0;
    SelFetch<SEL_T * SEL_B1> f1;
    f0.fetch(w.hist0);
    f1.fetch(w.hist1);
    const long long nz_now = threadIdx.x == 0 ? __hip_atomic_load(&st->nz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    __syncthreads();                                         // every thread is done with the unpadded h0
    f0.template store<true>(h0);
    f1.template store<false>(h1);
    __syncthreads();

    const int t = wv;
    const bool worker = t < SEL_SWAVES;
    unsigned long long n = 0, rank = 0;
    uint32_t prefix = 0;
    int row = -1;
    if (worker) {                                              // level 0 from the persistent histogram
        WaveBins<SEL_B0 / 64, SEL_PAD0> wb;
        wb.load(h0 + (t < SEL_T_RET ? 0 : SEL_H0_LDS / 2));
        n = wb.total;
        if (n > 0) {
            rank = sel_target_rank(t, n);
            prefix = (uint32_t)wb.find(rank);
            const int q0 = t < SEL_T_RET ? 0 : SEL_T_RET, q1 = t < SEL_T_RET ? ntrack_ret : SEL_T_RET + (ntrack - ntrack_ret);
            for (int q = q0; q < q1; ++q)
                if (tprefix[q] == prefix) row = tslot[q];
        }
        if (lane == 0) { s_rank[t] = rank; s_prefix[t] = prefix; s_row[t] = row; }
        if (threadIdx.x == 0) s_n = n;
    }
    __syncthreads();
    n = s_n;
    if (n == 0) {                                              // nothing finished yet: both full passes return at once
        if (threadIdx.x == 0) { st->n = 0; st->skip1 = 1; st->skip2 = 1; }
        return;
    }
    bool tracked = true;
    for (int u = 0; u < SEL_T; ++u) tracked = tracked && s_row[u] >= 0;
    if (!tracked) {
        // A wanted rank sits under a prefix without a persistent level-1 histogram: track the nine current
        // prefixes instead and let the level-1 pass rebuild their histograms from the rings.
        for (int k = threadIdx.x; k < SEL_T * SEL_B1; k += SEL_HBLOCK)
            __hip_atomic_store(&w.hist1[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) {
            sel_setup_next(st, s_prefix, s_rank, n, true);
            st->skip1 = 0;
            st->skip2 = 0;                                     // decided by the level-1 pass
        }
        return;
    }
    if (worker) {                                              // level 1 from the tracked histograms
        WaveBins<SEL_B1 / 64> wb;
        wb.load(h1 + row * SEL_B1);
        prefix = (prefix << 10) | (uint32_t)wb.find(rank);
    }
    __syncthreads();                                           // every wave has read s_rank / s_prefix of level 0
    if (worker && lane == 0) { s_rank[t] = rank; s_prefix[t] = prefix; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    st->skip1 = 1;
    if (nz_now == 0) {
        st->n = n;
        st->skip2 = 1;
        sel_finalize(stats5, s_prefix, 22, n);
    } else {                                                   // values with low float bits set: level 2 over the rings
        sel_setup_next(st, s_prefix, s_rank, n, false);
        st->skip2 = 0;
    }
}

// ---- incremental form for small integers: ONE launch, one level -------------------------------------
// What the env kernels of this library write into the rings are small integers: returns are sums of +-1 rewards under
// TimeLimit(500) (|return| <= 500), lengths step counts <= 500.  For such rings the window's values can be counted
// DIRECTLY -- one bin per integer -- instead of by key digits: returns in [-512, 512] into 1025 bins, lengths in
// [0, 512] into 513.  The persistent histogram is then the sorted window itself: nine ranks are nine positions in a
// prefix sum over ~1.5 k bins, there are no levels, no tracked prefixes and no pass that may or may not have work --
// k_sel_incr's two fallback launches (a kernel boundary each, ~4.5 us idle) are gone, and the LDS clear, the flush
// and the last workgroup's read-back shrink from 17 k words to 2 k.  The walk over the newly written slots and the
// shadow copy are k_sel_incr's.  A value that is not a small integer is counted in `bad` for as long as it stays in
// the window and turns the statistics into NaN (sel_poison): the promise is checked, not trusted.
constexpr int INTS_R0 = PPO_STATS_INTS_RET_ZERO;           // bin of return 0
constexpr int INTS_RBINS = PPO_STATS_INTS_RET_BINS;        // 1280 = 64 lanes x 20 (1025 used)
constexpr int INTS_LBINS = PPO_STATS_INTS_LEN_BINS;        // 768 = 64 lanes x 12 (513 used)
constexpr int INTS_WORDS = INTS_RBINS + INTS_LBINS;
// The persistent histogram is kept as INTS_REPL partial histograms whose u32 sum (mod 2^32: a partial count may have
// wrapped below zero) is the window's: a workgroup flushes into copy blockIdx % INTS_REPL.  With ONE copy the flush
// was 5.0 us of the launch's 16 -- 256 workgroups' atomics queueing on the same ~100 hot bins (returns and lengths of
// 8..60 for a fresh policy), tools/stats_timeline.py -- with eight it is 1.2 us; the last workgroup reads eight copies
// instead of one in the same round trip.
constexpr int INTS_REPL = PPO_STATS_INTS_REPLICAS;
static_assert(INTS_RBINS % 256 == 0 && INTS_LBINS % 256 == 0 && INTS_R0 * 2 + 1 <= INTS_RBINS, "bin layout");

struct IntsState {                // head of the state block (the first SEL_STATE_BYTES bytes), zero-filled by the caller
    unsigned long long n;         // valid window entries at the last call
    long long bad;                // window entries that are not small integers
    uint32_t ticket;
};
static_assert(sizeof(IntsState) <= SEL_STATE_BYTES, "state head too small");

__device__ __forceinline__ void ints_apply(uint32_t *h, uint32_t sign, float r, int l, int &bad)
{
    const int ri = (int)r;                                     // (saturates for huge values, which then fail the range test)
    const bool ok = (float)ri == r && ri >= -INTS_R0 && ri <= INTS_R0 && l >= 0 && l <= PPO_STATS_INTS_LEN_MAX;
    if (ok) {
        atomicAdd(&h[INTS_R0 + ri], sign);
        atomicAdd(&h[INTS_RBINS + l], sign);
    } else {
        bad += (int)sign;                                      // +1 entering, -1 (0xffffffff) leaving
    }
}

#ifdef PPO_STATS_TIMELINE                          // development (tools/stats_timeline.py): the launch's phases per workgroup
__device__ unsigned long long g_stats_timeline[256 * 8];
#define TLS_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_stats_timeline[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TLS_MARK(k) do { } while (0)
#endif
__global__ void __launch_bounds__(SEL_HBLOCK, 1) k_sel_incr_ints(const float *__restrict__ ret_ring, const int *__restrict__ len_ring,
                                                                const int *__restrict__ ring_cnt, int ring, int N,
                                                                IntsState *st, uint32_t *hist, IncrBufs ib, double *stats5)
{
    __shared__ __attribute__((aligned(16))) uint32_t h[INTS_WORDS];
    __shared__ int s_is_last;
    __shared__ int s_bin[SEL_T];
    __shared__ unsigned long long s_n;
    __shared__ uint32_t s_first[INCR_ENVS + 1];
    __shared__ int s_c0[INCR_ENVS], s_c1[INCR_ENVS];
    __shared__ uint32_t s_wsum[INCR_ENVS / 64];
    uint2 *__restrict__ shadow = ib.shadow;
    TLS_MARK(0);
    for (int k = threadIdx.x; k < INTS_WORDS; k += SEL_HBLOCK) h[k] = 0u;

    // the slots written since the last call, exactly as in k_sel_incr
    const uint32_t uring = (uint32_t)ring;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int bad = 0;
    for (int env0 = blockIdx.x * INCR_ENVS; env0 < N; env0 += gridDim.x * INCR_ENVS) {
        __syncthreads();
        uint32_t m = 0, incl = 0;
        int c0 = 0, c1 = 0;
        if (threadIdx.x < INCR_ENVS) {
            const int n = env0 + threadIdx.x;
            if (n < N) {
                c1 = max(ring_cnt[n], 0);
                c0 = ib.cnt_prev[n];
                m = (uint32_t)max(min(c1 - c0, ring), 0);
                if (m > 0) ib.cnt_prev[n] = c1;
            }
            incl = m;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off, 64);
                if (lane >= off) incl += up;
            }
            if (lane == 63) s_wsum[wv] = incl;
            s_c0[threadIdx.x] = c0;
            s_c1[threadIdx.x] = c1;
        }
        __syncthreads();
        if (threadIdx.x < INCR_ENVS) {
            uint32_t before = 0;
            for (int q = 0; q < wv; ++q) before += s_wsum[q];
            s_first[threadIdx.x] = before + incl - m;
            if (threadIdx.x == INCR_ENVS - 1) s_first[INCR_ENVS] = before + incl;
        }
        __syncthreads();
        TLS_MARK(1);
        const uint32_t items = s_first[INCR_ENVS];
        for (uint32_t it0 = threadIdx.x; it0 < items; it0 += 2 * SEL_HBLOCK) {
            size_t at[2];
            bool live[2], old_valid[2];
            float r_new[2], r_old[2];
            int l_new[2], l_old[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t it = it0 + u * SEL_HBLOCK;
                live[u] = it < items;
                int e = 0;
#pragma unroll
                for (int stepb = INCR_ENVS / 2; stepb > 0; stepb >>= 1)
                    if (s_first[e + stepb] <= it) e += stepb;
                if (!live[u]) e = 0;
                const uint32_t first = s_first[e], cnt_e = s_first[e + 1] - first;
                const int ec0 = s_c0[e], ec1 = s_c1[e];
                const uint32_t i = (uint32_t)ec1 - cnt_e + (it - first);
                const uint32_t slot = live[u] ? i % uring : 0u;
                at[u] = (size_t)(env0 + e) * uring + slot;
                old_valid[u] = live[u] && (ec0 >= ring || slot < (uint32_t)ec0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                r_new[u] = ret_ring[at[u]]; l_new[u] = len_ring[at[u]];
                const uint2 old = shadow[at[u]];
                r_old[u] = __uint_as_float(old.x); l_old[u] = (int)old.y;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (live[u]) {
                    shadow[at[u]] = make_uint2(__float_as_uint(r_new[u]), (uint32_t)l_new[u]);
                    ints_apply(h, 1u, r_new[u], l_new[u], bad);
                }
                if (old_valid[u]) ints_apply(h, 0xFFFFFFFFu, r_old[u], l_old[u], bad);
            }
        }
    }
    TLS_MARK(2);
    bad = (int)wave_sum((double)bad);
    if ((threadIdx.x & 63) == 0 && bad != 0)
        __hip_atomic_fetch_add(&st->bad, (long long)bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    uint32_t *mine = hist + (blockIdx.x % INTS_REPL) * INTS_WORDS;
    for (int k = threadIdx.x; k < INTS_WORDS; k += SEL_HBLOCK) {
        const uint32_t v = h[k];
        if (v) atomicAdd(&mine[k], v);
    }
    // last workgroup to flush makes the statistics (see k_sel for why atomics + vmcnt(0) and no fence)
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
    __syncthreads();
    TLS_MARK(3);
    if (threadIdx.x == 0)
        s_is_last = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    TLS_MARK(4);
    if (!s_is_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SelFetch<INTS_WORDS> f;                                  // the sum of the copies, every load in flight at once
    {
        SelFetch<INTS_WORDS> part[INTS_REPL];
#pragma unroll
        for (int c = 0; c < INTS_REPL; ++c) part[c].fetch(hist + c * INTS_WORDS);
#pragma unroll
        for (int i = 0; i < SelFetch<INTS_WORDS>::PER; ++i) {
            uint32_t lo = 0u, hi = 0u;
#pragma unroll
            for (int c = 0; c < INTS_REPL; ++c) { lo += (uint32_t)part[c].v[i]; hi += (uint32_t)(part[c].v[i] >> 32); }
            f.v[i] = (unsigned long long)lo | ((unsigned long long)hi << 32);
        }
    }
    const long long bad_now = threadIdx.x == 0 ? __hip_atomic_load(&st->bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    __syncthreads();                                         // every thread is done with the deltas
    f.template store<false>(h);
    __syncthreads();
    TLS_MARK(5);
    const int t = wv;                                        // waves 0..8: one target each
    if (t < SEL_SWAVES) {
        unsigned long long n = 0, rank = 0;
        int bin = 0;
        if (t < SEL_T_RET) {
            WaveBins<INTS_RBINS / 64> wb;
            wb.load(h);
            n = wb.total;
            if (n > 0) { rank = sel_target_rank(t, n); bin = wb.find(rank); }
        } else {
            WaveBins<INTS_LBINS / 64> wb;
            wb.load(h + INTS_RBINS);
            n = wb.total;
            if (n > 0) { rank = sel_target_rank(t, n); bin = wb.find(rank); }
        }
        if (lane == 0) s_bin[t] = bin;
        if (threadIdx.x == 0) s_n = n;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const unsigned long long n = s_n;
    st->n = n;
    if (bad_now != 0) { sel_poison(stats5); return; }
    if (n == 0) return;                                      // nothing finished yet: stats5 keeps its value (NaN at start)
    double v[SEL_T];
    for (int u = 0; u < SEL_T; ++u) v[u] = u < SEL_T_RET ? (double)(s_bin[u] - INTS_R0) : (double)s_bin[u];
    if (!(stats5[0] >= v[6])) stats5[0] = v[6];              // also replaces the initial NaN
    stats5[1] = lerp_pair(v[0], v[1], n, 0.05);
    stats5[2] = lerp_pair(v[2], v[3], n, 0.50);
    stats5[3] = lerp_pair(v[4], v[5], n, 0.95);
    stats5[4] = lerp_pair(v[7], v[8], n, 0.50);
    TLS_MARK(6);
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

size_t ppo_episode_stats_hist0_offset(void) { return SEL_STATE_BYTES; }

int ppo_episode_hist_level(const float *ret_ring, const int *len_ring, const int *ring_cnt, int ring, int N, int level,
                           const uint32_t *prefixes, int n_ret, int n_len, uint32_t *hist, void *stream)
{
    PPO_REQUIRE(N >= 0 && ring > 0 && (unsigned long long)N * (unsigned long long)ring < 0xFFFFFFFFull);
    PPO_REQUIRE((level == 1 || level == 2) && prefixes && hist && ppo_aligned(prefixes, 4) && ppo_aligned(hist, 4));
    PPO_REQUIRE(n_ret >= 0 && n_ret <= SEL_T_RET && n_len >= 0 && n_len <= SEL_T - SEL_T_RET);
    if (N == 0) return 0;
    PPO_REQUIRE(ret_ring && len_ring && ring_cnt);
    size_t blocks = ((size_t)N * ring + SEL_HBLOCK * SEL_U - 1) / (SEL_HBLOCK * SEL_U);
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    hipStream_t s = (hipStream_t)stream;
    if (level == 1)
        PPO_LAUNCH(k_hist_level<1>, dim3((unsigned)blocks), dim3(SEL_HBLOCK), s, ret_ring, len_ring, ring_cnt, ring, N, prefixes,
                   n_ret, n_len, hist);
    else
        PPO_LAUNCH(k_hist_level<2>, dim3((unsigned)blocks), dim3(SEL_HBLOCK), s, ret_ring, len_ring, ring_cnt, ring, N, prefixes,
                   n_ret, n_len, hist);
    return ppo_launch_status();
}

size_t ppo_episode_stats_state_bytes(int N, int ring)
{
    if (N < 0 || ring <= 0) return 0;
    return incr_state_bytes(N, ring);
}

int ppo_episode_stats_incremental(const float *ret_ring, const int *len_ring, const int *ring_cnt, int ring, int N,
                                  double *stats5, void *state, size_t state_bytes, void *stream)
{
    PPO_REQUIRE(N >= 0 && ring > 0 && (unsigned long long)N * (unsigned long long)ring < 0xFFFFFFFFull);
    PPO_REQUIRE(stats5 && ppo_aligned(stats5, 8));
    PPO_REQUIRE(state && state_bytes >= incr_state_bytes(N, ring) && ppo_aligned(state, 16));
    if (N == 0) return 0;
    PPO_REQUIRE(ret_ring && len_ring && ring_cnt);
    hipStream_t s = (hipStream_t)stream;
    const SelWs w = sel_ws(state);
    char *p = static_cast<char *>(state) + INCR_HEAD_BYTES;
    IncrBufs ib;
    ib.cnt_prev = reinterpret_cast<int *>(p);
    ib.shadow = reinterpret_cast<uint2 *>(p + incr_cnt_bytes(N));
    size_t iblocks = ((size_t)N + INCR_ENVS - 1) / INCR_ENVS;
    if (iblocks > 256) iblocks = 256;
    PPO_LAUNCH(k_sel_incr, dim3((unsigned)iblocks), dim3(SEL_HBLOCK), s, ret_ring, len_ring, ring_cnt, ring, N, w,
                       ib, stats5);
    int rc = ppo_launch_status();
    if (rc) return rc;
    // the two full passes: return at once unless a wanted prefix is untracked (level 1) or values are
    // not integers (level 2)
    // (one workgroup per CU instead of two: what these launches cost when they have nothing to do is
    // the dispatch of their waves)
    size_t blocks = ((size_t)N * ring + SEL_HBLOCK * SEL_U - 1) / (SEL_HBLOCK * SEL_U);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    rc = sel_level<1, true>(ret_ring, len_ring, ring_cnt, ring, N, w, stats5, (unsigned)blocks, s);
    if (rc) return rc;
    return sel_level<2, true>(ret_ring, len_ring, ring_cnt, ring, N, w, stats5, (unsigned)blocks, s);
}

size_t ppo_episode_stats_ints_hist_offset(void) { return SEL_STATE_BYTES; }

int ppo_episode_stats_incremental_ints(const float *ret_ring, const int *len_ring, const int *ring_cnt, int ring, int N,
                                       double *stats5, void *state, size_t state_bytes, void *stream)
{
    PPO_REQUIRE(N >= 0 && ring > 0 && (unsigned long long)N * (unsigned long long)ring < 0xFFFFFFFFull);
    PPO_REQUIRE(stats5 && ppo_aligned(stats5, 8));
    PPO_REQUIRE(state && state_bytes >= incr_state_bytes(N, ring) && ppo_aligned(state, 16));
    static_assert(SEL_STATE_BYTES + INTS_REPL * INTS_WORDS * sizeof(uint32_t) <= INCR_HEAD_BYTES, "direct histograms fit the state head");
    if (N == 0) return 0;
    PPO_REQUIRE(ret_ring && len_ring && ring_cnt);
    char *base = static_cast<char *>(state);
    IntsState *st = reinterpret_cast<IntsState *>(base);
    uint32_t *hist = reinterpret_cast<uint32_t *>(base + SEL_STATE_BYTES);
    char *p = base + INCR_HEAD_BYTES;
    IncrBufs ib;
    ib.cnt_prev = reinterpret_cast<int *>(p);
    ib.shadow = reinterpret_cast<uint2 *>(p + incr_cnt_bytes(N));
    size_t iblocks = ((size_t)N + INCR_ENVS - 1) / INCR_ENVS;
    if (iblocks > 256) iblocks = 256;
    PPO_LAUNCH(k_sel_incr_ints, dim3((unsigned)iblocks), dim3(SEL_HBLOCK), (hipStream_t)stream, ret_ring, len_ring, ring_cnt,
               ring, N, st, hist, ib, stats5);
    return ppo_launch_status();
}

}  // extern "C"

#ifdef PPO_STATS_TIMELINE
extern "C" int ppo_stats_timeline_read(unsigned long long *dst, int words)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stats_timeline), (size_t)words * 8, 0, hipMemcpyDeviceToHost);
}
#endif
