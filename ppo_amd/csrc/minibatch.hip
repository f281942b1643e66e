// Minibatch gather for the update phase -- gfx950.
//
// Replaces the five fancy-index gathers of reference src/ppo.py:582-588
// (states[idx], actions[idx], log_probs[idx], values[idx], advantages[idx]) with one
// launch.  The flattened rollout (T*N rows) is read at random rows; at C2 scale the
// scalar arrays are 33.5 MB each and the observations 134 MB, all inside the 256 MiB
// Infinity Cache right after the rollout wrote them, so the gather runs at cache, not
// HBM, latency.  One lane per sample: the 8-byte index load and every store are
// coalesced; observation / action rows move as one 16-byte access when they are 4 words
// wide and aligned, word by word otherwise.  Pure data movement: bit-exact by definition.
//
// k_shuffle_indices: the permutation itself (reference src/ppo.py:575-576, np.random.shuffle(indices) once
// per epoch).  torch.randperm sorts 64-bit random keys: eight radix passes over 8.4 M pairs, 0.59 ms per
// epoch at BASELINE configs[1] -- more than a minibatch step of the update.  Here index i is mapped through
// a keyed bijection of [0, 2^k), 2^k >= n (xorshift / odd-multiply / add rounds: each step is invertible
// modulo 2^k, so the whole is), applied again while the result is >= n (cycle walking: still a
// bijection of [0, n), fewer than two applications on average).  One pass, 8 B written per index, no sort.
// The keys are Philox4x32-10 words of (seed, stream): integer arithmetic only, the oracle restates it bit
// for bit (oracle/ppo_oracle.c: oracle_shuffle_indices).
#include "common.h"

namespace {

constexpr int MB_BLOCK = 256;

struct GatherArgs {
    const int64_t *idx;
    const uint32_t *states, *actions;  // rows of state_words / action_words 32-bit words
    const float *log_probs, *values, *adv;
    uint32_t *mb_states, *mb_actions;
    float *mb_log_probs, *mb_values, *mb_adv;
    int *err_flag;
    long B, M;
    int state_words, action_words;
    bool state_vec4, action_vec4;
};

__device__ __forceinline__ void gather_row(const uint32_t *src, uint32_t *dst, long from, long to, int words,
                                           bool vec4, bool ok)
{
    if (vec4) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (ok) v = reinterpret_cast<const uint4 *>(src)[from];
        reinterpret_cast<uint4 *>(dst)[to] = v;
    } else {
        for (int w = 0; w < words; ++w) dst[to * words + w] = ok ? src[from * words + w] : 0u;
    }
}

__global__ void __launch_bounds__(MB_BLOCK) k_minibatch_gather(GatherArgs a)
{
    for (long i = (long)blockIdx.x * MB_BLOCK + threadIdx.x; i < a.B; i += (long)gridDim.x * MB_BLOCK) {
        long j = a.idx[i];
        if (j < 0) j += a.M;                 // torch / numpy negative-index wrap
        const bool ok = j >= 0 && j < a.M;   // torch raises IndexError; here: zeros + err_flag
        if (!ok) {
            if (a.err_flag) *a.err_flag = 1;
            j = 0;
        }
        if (a.states) gather_row(a.states, a.mb_states, j, i, a.state_words, a.state_vec4, ok);
        if (a.actions) gather_row(a.actions, a.mb_actions, j, i, a.action_words, a.action_vec4, ok);
        if (a.log_probs) a.mb_log_probs[i] = ok ? a.log_probs[j] : 0.0f;
        if (a.values) a.mb_values[i] = ok ? a.values[j] : 0.0f;
        if (a.adv) a.mb_adv[i] = ok ? a.adv[j] : 0.0f;
    }
}

// Wide rows (image observations: 84 x 84 x 4 f32 = 113 KB per sample, BASELINE configs[3]): one lane
// per 16-byte piece of a row instead of one lane per sample, so that a wave moves 1 KiB of one row
// per access instead of 64 words of 64 different rows.
__global__ void __launch_bounds__(MB_BLOCK) k_gather_wide_rows(const int64_t *idx, const uint4 *src, uint4 *dst, long B,
                                                               long M, long vec_per_row, int *err_flag)
{
    const long total = B * vec_per_row;
    for (long e = (long)blockIdx.x * MB_BLOCK + threadIdx.x; e < total; e += (long)gridDim.x * MB_BLOCK) {
        const long i = e / vec_per_row, w = e - i * vec_per_row;
        long j = idx[i];
        if (j < 0) j += M;
        const bool ok = j >= 0 && j < M;
        if (!ok && err_flag && w == 0) *err_flag = 1;
        dst[e] = ok ? src[j * vec_per_row + w] : make_uint4(0u, 0u, 0u, 0u);
    }
}

constexpr int WIDE_ROW_WORDS = 32;   // rows at least this wide (and 16-byte granular) take the wide-row kernel

constexpr int SHUF_ROUNDS = 6;
struct ShuffleKeys {
    uint32_t mul[SHUF_ROUNDS], add[SHUF_ROUNDS];   // odd multipliers, additive keys
    uint32_t mask;                                 // 2^k - 1
    int sh_a, sh_b;                                // the two xorshift distances: ceil(k / 2), max(1, k / 3)
};

__host__ __device__ inline uint32_t shuffle_bijection(uint32_t x, const ShuffleKeys &q)
{
#pragma unroll
    for (int r = 0; r < SHUF_ROUNDS; ++r) {
        x ^= x >> ((r & 1) ? q.sh_b : q.sh_a);
        x = (x * q.mul[r] + q.add[r]) & q.mask;
    }
    x ^= x >> q.sh_a;
    return x;
}

__global__ void __launch_bounds__(MB_BLOCK) k_shuffle_indices(ShuffleKeys q, long n, int64_t *out)
{
    for (long i = (long)blockIdx.x * MB_BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * MB_BLOCK) {
        uint32_t x = (uint32_t)i;
        do x = shuffle_bijection(x, q);
        while ((long)x >= n);                      // cycle walking: i < n is on the cycle, so this ends
        out[i] = (int64_t)x;
    }
}

// Philox4x32-10 on the host (common.h has the device form): the round keys are made once per launch
void host_philox4x32_10(const uint32_t c_in[4], uint32_t k0, uint32_t k1, uint32_t out[4])
{
    uint32_t c0 = c_in[0], c1 = c_in[1], c2 = c_in[2], c3 = c_in[3];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// The keys of the permutation of [0, n) that (seed, stream_id) select.
ShuffleKeys make_shuffle_keys(long n, uint64_t seed, uint64_t stream_id)
{
    ShuffleKeys q;
    int k = 2;
    while (k < 32 && (1L << k) < n) ++k;
    q.mask = k == 32 ? 0xFFFFFFFFu : ((1u << k) - 1u);
    q.sh_a = (k + 1) / 2;
    q.sh_b = k / 3 > 1 ? k / 3 : 1;
    // key half 1 is offset so that this stream never collides with the reset / sampling streams of the same seed
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ 0x2545F491u;
    uint32_t w[12];
    for (uint32_t b = 0; b < 3; ++b) {
        const uint32_t ctr[4] = {(uint32_t)stream_id, (uint32_t)(stream_id >> 32), b, 0x5348464Cu};
        host_philox4x32_10(ctr, k0, k1, w + 4 * b);
    }
    for (int r = 0; r < SHUF_ROUNDS; ++r) {
        q.mul[r] = w[r] | 1u;
        q.add[r] = w[SHUF_ROUNDS + r];
    }
    return q;
}

// ---- advantage statistics of every minibatch of an epoch in one streaming pass -------------------
// The minibatches of an epoch are slices of the shuffled indices (reference src/ppo.py:575-588), and
// compute_losses normalises each minibatch's advantages with its own mean / std (:501).  Gathering
// adv[idx] per minibatch reads a 64-byte sector for every 4-byte value (43 us per 2 M-sample minibatch at
// BASELINE configs[1], four times per epoch).  The permutation is a keyed bijection, so it can be run
// BACKWARDS instead: element j of the flat rollout sits at position inv(j) of the shuffled order, i.e. in
// minibatch inv(j) / batch -- one coalesced pass over adv serves all minibatches of the epoch at once.
// Each step of shuffle_bijection is inverted exactly: x ^= x >> s by the doubling series, the affine step
// by the multiplier's inverse modulo 2^k (Newton's iteration on the host); cycle walking works the same
// way in reverse.  Branch-free per application (the doubling steps beyond k bits are masked off by host-made masks; the
// minibatch of a position is a count of compares against host-made bounds).  Shifted sums as in loss.hip (K = adv[0]);
// per-thread f64 accumulators {sum, sum of squares} live in LDS as one 16-byte word per (minibatch, thread), blocks and
// threads add in a fixed order.
constexpr int EPOCH_MAX_MB = 8;
constexpr int EPOCH_BLOCK = 256;
constexpr int EPOCH_GRID = 1024;

struct ShuffleInvKeys {
    uint32_t inv[SHUF_ROUNDS], add[SHUF_ROUNDS];
    uint32_t mask;
    int sh_a;                      // 2 sh_a >= k: one step undoes x ^= x >> sh_a
    int sb[3];                     // sh_b, 2 sh_b, 4 sh_b (clamped to 31): the doubling series of the undo ...
    uint32_t mb[3];                // ... and all-ones / zero for the steps below / beyond k bits (branch-free)
    unsigned long long bound[EPOCH_MAX_MB - 1];   // first position of minibatch m + 1 (~0 for minibatches that do not exist)
};

__device__ __forceinline__ uint32_t shuffle_bijection_inv(uint32_t y, const ShuffleInvKeys &q)
{
    y ^= y >> q.sh_a;
#pragma unroll
    for (int r = SHUF_ROUNDS - 1; r >= 0; --r) {
        y = ((y - q.add[r]) * q.inv[r]) & q.mask;
        if (r & 1) {
            y ^= y >> q.sb[0];
            y ^= (y >> q.sb[1]) & q.mb[1];
            y ^= (y >> q.sb[2]) & q.mb[2];
        } else {
            y ^= y >> q.sh_a;
        }
    }
    return y;
}

__global__ void __launch_bounds__(EPOCH_BLOCK) k_adv_stats_epoch(ShuffleInvKeys q, const float *adv, long n, int num_mb,
                                                                  double *partials)
{
    // per-thread accumulators {sum, sum of squares} of every minibatch: one 16-byte LDS word each
    __shared__ double2 slot[EPOCH_MAX_MB][EPOCH_BLOCK];
    const int tid = threadIdx.x;
    for (int c = 0; c < num_mb; ++c) slot[c][tid] = make_double2(0.0, 0.0);
    const double K = (double)adv[0];
    const long stride = (long)gridDim.x * EPOCH_BLOCK;
    for (long j0 = (long)blockIdx.x * EPOCH_BLOCK + tid; j0 < n; j0 += 4 * stride) {
        float e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = j0 + u * stride < n ? adv[j0 + u * stride] : 0.0f;
        int mb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long j = j0 + u * stride;
            uint32_t x = (uint32_t)(j < n ? j : 0);
            do x = shuffle_bijection_inv(x, q);
            while ((long)x >= n);                      // cycle walking, backwards
            int m = 0;
#pragma unroll
            for (int b = 0; b < EPOCH_MAX_MB - 1; ++b) m += (unsigned long long)x >= q.bound[b];
            mb[u] = j < n ? m : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (mb[u] >= 0) {
                const double d = (double)e[u] - K;
                double2 s = slot[mb[u]][tid];
                s.x += d;
                s.y += d * d;
                slot[mb[u]][tid] = s;
            }
        }
    }
    // this block's sums: every wave reduces its threads' slots, thread 0 adds the waves in order
    __shared__ double part[EPOCH_BLOCK / PPO_WAVE][EPOCH_MAX_MB * 2];
    for (int c = 0; c < num_mb; ++c) {
        const double2 s = slot[c][tid];
        const double w0 = wave_sum(s.x), w1 = wave_sum(s.y);
        if ((tid & (PPO_WAVE - 1)) == 0) { part[tid / PPO_WAVE][2 * c] = w0; part[tid / PPO_WAVE][2 * c + 1] = w1; }
    }
    __syncthreads();
    if (tid < num_mb * 2) {
        double t = 0.0;
        for (int w = 0; w < EPOCH_BLOCK / PPO_WAVE; ++w) t += part[w][tid];
        partials[(size_t)blockIdx.x * (EPOCH_MAX_MB * 2) + tid] = t;
    }
}

__global__ void __launch_bounds__(EPOCH_BLOCK) k_adv_stats_epoch_final(const float *adv, long n, long batch, int num_mb,
                                                                        const double *partials, int nblocks, double *stats3)
{
    // thread = (row group g, column c): column c of the rows g, g + 16, ... (coalesced: 16 consecutive doubles per row),
    // then the 16 groups are added in order
    constexpr int NC = EPOCH_MAX_MB * 2, NG = EPOCH_BLOCK / NC;
    __shared__ double part[NG][NC];
    const int tid = threadIdx.x, c = tid % NC, g = tid / NC;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;                 // four independent chains: loads in flight
    int b = g;
    for (; b + 3 * NG < nblocks; b += 4 * NG) {
        a0 += partials[(size_t)b * NC + c];
        a1 += partials[(size_t)(b + NG) * NC + c];
        a2 += partials[(size_t)(b + 2 * NG) * NC + c];
        a3 += partials[(size_t)(b + 3 * NG) * NC + c];
    }
    for (; b < nblocks; b += NG) a0 += partials[(size_t)b * NC + c];
    part[g][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (tid < num_mb) {
        double S = 0.0, Q = 0.0;
        for (int w = 0; w < NG; ++w) { S += part[w][2 * tid]; Q += part[w][2 * tid + 1]; }
        const long lo = tid * batch, hi = lo + batch < n ? lo + batch : n;
        const double cnt = (double)(hi - lo), K = (double)adv[0];
        stats3[3 * tid + 0] = cnt;
        stats3[3 * tid + 1] = K + S / cnt;
        const double m2 = Q - S * S / cnt;
        stats3[3 * tid + 2] = m2 > 0.0 ? m2 : 0.0;
    }
}

}  // namespace

extern "C" {

int ppo_minibatch_gather(const int64_t *idx, long B, long M, const void *states, int state_words,
                         const void *actions, int action_words, const float *log_probs, const float *values,
                         const float *adv, void *mb_states, void *mb_actions, float *mb_log_probs,
                         float *mb_values, float *mb_adv, int *err_flag, void *stream)
{
    PPO_REQUIRE(idx && B >= 1 && M >= 1);
    PPO_REQUIRE(!states || (mb_states && state_words >= 1));
    PPO_REQUIRE(!actions || (mb_actions && action_words >= 1));
    PPO_REQUIRE(!log_probs || mb_log_probs);
    PPO_REQUIRE(!values || mb_values);
    PPO_REQUIRE(!adv || mb_adv);
    PPO_REQUIRE(ppo_aligned(idx, 8) && ppo_aligned(states, 4) && ppo_aligned(actions, 4) &&
                ppo_aligned(mb_states, 4) && ppo_aligned(mb_actions, 4));
    GatherArgs a;
    a.idx = idx;
    a.states = static_cast<const uint32_t *>(states);
    a.actions = static_cast<const uint32_t *>(actions);
    a.log_probs = log_probs; a.values = values; a.adv = adv;
    a.mb_states = static_cast<uint32_t *>(mb_states);
    a.mb_actions = static_cast<uint32_t *>(mb_actions);
    a.mb_log_probs = mb_log_probs; a.mb_values = mb_values; a.mb_adv = mb_adv;
    a.err_flag = err_flag;
    a.B = B; a.M = M; a.state_words = state_words; a.action_words = action_words;
    hipStream_t s = (hipStream_t)stream;
    if (states && state_words >= WIDE_ROW_WORDS && state_words % 4 == 0 && ppo_aligned(states, 16) &&
        ppo_aligned(mb_states, 16)) {
        const long vpr = state_words / 4;
        long wb = (B * vpr + MB_BLOCK - 1) / MB_BLOCK;
        if (wb > 16384) wb = 16384;
        hipLaunchKernelGGL(k_gather_wide_rows, dim3((unsigned)wb), dim3(MB_BLOCK), 0, s, idx,
                           static_cast<const uint4 *>(states), static_cast<uint4 *>(mb_states), B, M, vpr, err_flag);
        const int rc = ppo_launch_status();
        if (rc) return rc;
        a.states = nullptr;                          // the per-sample kernel below moves the narrow arrays only
    }
    a.state_vec4 = states && state_words == 4 && ppo_aligned(states, 16) && ppo_aligned(mb_states, 16);
    a.action_vec4 = actions && action_words == 4 && ppo_aligned(actions, 16) && ppo_aligned(mb_actions, 16);
    long blocks = (B + MB_BLOCK - 1) / MB_BLOCK;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_minibatch_gather, dim3((unsigned)blocks), dim3(MB_BLOCK), 0, s, a);
    return ppo_launch_status();
}

int ppo_shuffle_indices(long n, uint64_t seed, uint64_t stream_id, int64_t *idx_out, void *stream)
{
    PPO_REQUIRE(n >= 0 && n <= (1L << 32));
    if (n == 0) return 0;
    PPO_REQUIRE(idx_out && ppo_aligned(idx_out, 8));
    const ShuffleKeys q = make_shuffle_keys(n, seed, stream_id);
    long blocks = (n + MB_BLOCK - 1) / MB_BLOCK;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_shuffle_indices, dim3((unsigned)blocks), dim3(MB_BLOCK), 0, (hipStream_t)stream, q, n, idx_out);
    return ppo_launch_status();
}

size_t ppo_adv_stats_epoch_workspace_bytes(void) { return (size_t)EPOCH_GRID * EPOCH_MAX_MB * 2 * sizeof(double); }

int ppo_adv_stats_epoch(const float *adv, long n, uint64_t seed, uint64_t stream_id, long batch, double *stats3,
                        void *workspace, size_t workspace_bytes, void *stream)
{
    PPO_REQUIRE(n >= 1 && n <= (1L << 32) && batch >= 1);
    const long num_mb = (n + batch - 1) / batch;
    PPO_REQUIRE(num_mb <= EPOCH_MAX_MB);
    PPO_REQUIRE(adv && stats3 && workspace && ppo_aligned(stats3, 8) && ppo_aligned(workspace, 8));
    PPO_REQUIRE(workspace_bytes >= ppo_adv_stats_epoch_workspace_bytes());
    const ShuffleKeys f = make_shuffle_keys(n, seed, stream_id);
    ShuffleInvKeys q;
    q.mask = f.mask; q.sh_a = f.sh_a;
    int k = 2;
    while (k < 32 && (1L << k) < n) ++k;
    for (int i = 0; i < 3; ++i) {
        const int t = f.sh_b << i;
        q.sb[i] = t < 31 ? t : 31;
        q.mb[i] = t < k ? 0xFFFFFFFFu : 0u;
    }
    PPO_REQUIRE((f.sh_b << 3) >= k && 2 * f.sh_a >= k);   // three doubling steps (one for sh_a) undo the xorshifts for every k <= 32
    for (int r = 0; r < SHUF_ROUNDS; ++r) {
        uint32_t inv = f.mul[r];                               // odd a: a * a = 1 mod 8, each Newton step doubles the bits
        for (int it = 0; it < 5; ++it) inv *= 2u - f.mul[r] * inv;
        q.inv[r] = inv;
        q.add[r] = f.add[r];
    }
    for (int m = 0; m < EPOCH_MAX_MB - 1; ++m)
        q.bound[m] = m + 1 < num_mb ? (unsigned long long)(m + 1) * (unsigned long long)batch : ~0ull;
    long blocks = (n + EPOCH_BLOCK * 4 - 1) / (EPOCH_BLOCK * 4);
    if (blocks > EPOCH_GRID) blocks = EPOCH_GRID;
    hipStream_t s = (hipStream_t)stream;
    double *partials = static_cast<double *>(workspace);
    hipLaunchKernelGGL(k_adv_stats_epoch, dim3((unsigned)blocks), dim3(EPOCH_BLOCK), 0, s, q, adv, n, (int)num_mb, partials);
    int rc = ppo_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(k_adv_stats_epoch_final, dim3(1), dim3(EPOCH_BLOCK), 0, s, adv, n, batch, (int)num_mb,
                       (const double *)partials, (int)blocks, stats3);
    return ppo_launch_status();
}

}  // extern "C"
