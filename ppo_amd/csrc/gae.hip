// GAE(lambda) reverse-time scan over the [T, N] rollout buffer -- gfx950.
//
// Replaces PPO.compute_advantages (reference src/ppo.py:452-488).  HBM-bound: 20 B per
// (t, n) element (r, v, done, trunc read; adv written) + the end value where trunc != 0
// (24 B if the dense end-value buffer is streamed), ~6 flops.  No MFMA.
//
// Mapping: one lane owns VEC adjacent env columns and walks time backwards; consecutive
// lanes own consecutive columns, so every wave-level load/store is one contiguous
// 64*4*VEC-byte row segment (fully coalesced).  The recurrence itself is two dependent
// f32 ops per step; what has to be engineered is memory-level parallelism, because a
// [128, 65536] buffer is only 1024 waves wide (4 per CU).  Each lane therefore keeps TWO
// blocks of U rows x 5 arrays in flight in registers (software double buffering): the
// loads of block k+1 are issued before block k is consumed.
//
// Arithmetic: exactly the reference's sequence of f32 operations, one rounding each
// (the library is built with -ffp-contract=off), so results are bit-identical to
// torch-CPU for flags in {0,1} and reproduce its (1 - terminations) arithmetic otherwise.
#include "common.h"

PpoProf g_ppo_prof;   // bench-only launch timing, see common.h

namespace {

template <int VEC> struct VecT;
template <> struct VecT<1> { using F = float; using I = int; };
template <> struct VecT<2> { using F = float2; using I = int2; };
template <> struct VecT<4> { using F = float4; using I = int4; };

template <int VEC> __device__ __forceinline__ float fget(const typename VecT<VEC>::F &v, int k);
template <> __device__ __forceinline__ float fget<1>(const float &v, int) { return v; }
template <> __device__ __forceinline__ float fget<2>(const float2 &v, int k) { return k == 0 ? v.x : v.y; }
template <> __device__ __forceinline__ float fget<4>(const float4 &v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
template <int VEC> __device__ __forceinline__ int iget(const typename VecT<VEC>::I &v, int k);
template <> __device__ __forceinline__ int iget<1>(const int &v, int) { return v; }
template <> __device__ __forceinline__ int iget<2>(const int2 &v, int k) { return k == 0 ? v.x : v.y; }
template <> __device__ __forceinline__ int iget<4>(const int4 &v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
template <int VEC> __device__ __forceinline__ void fset(typename VecT<VEC>::F &v, int k, float x);
template <> __device__ __forceinline__ void fset<1>(float &v, int, float x) { v = x; }
template <> __device__ __forceinline__ void fset<2>(float2 &v, int k, float x) { if (k == 0) v.x = x; else v.y = x; }
template <> __device__ __forceinline__ void fset<4>(float4 &v, int k, float x) { if (k == 0) v.x = x; else if (k == 1) v.y = x; else if (k == 2) v.z = x; else v.w = x; }

// (DENSE_EV is always true since round 3: the ragged form has a kernel of its own, k_gae_ragged)
template <int VEC, int U, bool DENSE_EV> struct RowBlock {
    static_assert(DENSE_EV, "ragged end values: k_gae_ragged");
    typename VecT<VEC>::F r[U], v[U], ev[U];
    typename VecT<VEC>::I d[U], tr[U];
};

struct GaeArgs {
    const float *r, *v, *ev;   // ev: dense [T,N] or ragged [E]
    const int *done, *trunc;
    const uint32_t *flag_words; // packed forms only: [ceil(T/16), N], two bits per element (written by k_rollout)
    float *adv, *ret;
    const long *idx0;          // ragged only: start index per column (cumsum - 1)
    long E;
    int *err;
    float g, gl;
    int T, N;
};

template <int VEC, int U, bool DENSE_EV>
__device__ __forceinline__ void load_block(RowBlock<VEC, U, DENSE_EV> &b, const GaeArgs &a, size_t col, int t_top)
{
    using F = typename VecT<VEC>::F;
    using I = typename VecT<VEC>::I;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t o = (size_t)(t_top - u) * (size_t)a.N + col;
        b.r[u] = *reinterpret_cast<const F *>(a.r + o);
        b.v[u] = *reinterpret_cast<const F *>(a.v + o);
        b.d[u] = *reinterpret_cast<const I *>(a.done + o);
        b.tr[u] = *reinterpret_cast<const I *>(a.trunc + o);
        b.ev[u] = *reinterpret_cast<const F *>(a.ev + o);
    }
}

// One reverse step of one column: reference src/ppo.py:476-486.
__device__ __forceinline__ float gae_step(float r, float v, int d, int tr, float ev, float g, float gl,
                                          float &nv, float &na)
{
    const int term = d + tr;                 // :476
    const float keep = (float)(1 - term);
    nv = keep * nv + (float)tr * ev;         // :478
    na = keep * na;                          // :479
    const float delta = r + g * nv - v;      // :482
    const float adv = delta + gl * na;       // :483
    nv = v;                                  // :485
    na = adv;                                // :486
    return adv;
}

template <int VEC, int U, bool DENSE_EV>
__device__ __forceinline__ void consume_block(const RowBlock<VEC, U, DENSE_EV> &b, const GaeArgs &a, size_t col,
                                              int t_top, float (&nv)[VEC], float (&na)[VEC])
{
    using F = typename VecT<VEC>::F;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        F out, rout;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int d = iget<VEC>(b.d[u], k), tr = iget<VEC>(b.tr[u], k);
            const float ev = tr ? fget<VEC>(b.ev[u], k) : 0.0f;  // buffer is only defined where flagged
            const float v = fget<VEC>(b.v[u], k);
            const float adv = gae_step(fget<VEC>(b.r[u], k), v, d, tr, ev, a.g, a.gl, nv[k], na[k]);
            fset<VEC>(out, k, adv);
            fset<VEC>(rout, k, adv + v);                  // returns, src/ppo.py:498
        }
        const size_t o = (size_t)(t_top - u) * (size_t)a.N + col;
        *reinterpret_cast<F *>(a.adv + o) = out;
        if (a.ret) *reinterpret_cast<F *>(a.ret + o) = rout;
    }
}

template <int VEC, int U, bool DENSE_EV>
__global__ void __launch_bounds__(256) k_gae(GaeArgs a)
{
    const size_t col = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (col >= (size_t)a.N) return;

    float nv[VEC], na[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { nv[k] = 0.0f; na[k] = 0.0f; }   // :470-471

    const int nblk = a.T / U;
    int t = a.T - 1;
    RowBlock<VEC, U, DENSE_EV> cur, nxt;
    if (nblk > 0) load_block<VEC, U, DENSE_EV>(cur, a, col, t);
    for (int b = 0; b < nblk; ++b) {
        if (b + 1 < nblk) load_block<VEC, U, DENSE_EV>(nxt, a, col, t - U);  // prefetch next block
        consume_block<VEC, U, DENSE_EV>(cur, a, col, t, nv, na);
        cur = nxt;
        t -= U;
    }
    for (; t >= 0; --t) {                                 // T % U leftover rows
        RowBlock<VEC, 1, DENSE_EV> one;
        load_block<VEC, 1, DENSE_EV>(one, a, col, t);
        consume_block<VEC, 1, DENSE_EV>(one, a, col, t, nv, na);
    }
}

// ---------------------------------------------------------------------------------------
// Sparse-bootstrap form of the column-per-lane kernel.  end_value_dense is only ever used
// where trunc != 0 (one forced row per rollout plus the rare TimeLimit hits), so streaming
// it costs 4 of the 24 bytes per element for nothing.  Here the r/v/done/trunc loads run two
// blocks ahead, the flags of the NEXT block (already landed) decide -- per lane -- whether
// its end-value rows are fetched at all, and those few loads are in flight while the
// CURRENT block is consumed.  HBM traffic drops to 20 B per element plus ~one 64-byte
// sector per truncation.
template <int VEC, int U> struct FlagBlock {
    typename VecT<VEC>::F r[U], v[U];
    typename VecT<VEC>::I d[U], tr[U];
};
template <int VEC, int U> struct EvBlock {
    typename VecT<VEC>::F ev[U];
};

// NT: non-temporal accesses (every element is touched exactly once)
template <int BYTES> struct RawT;
template <> struct RawT<4> { using T = uint32_t; };
template <> struct RawT<8> { using T = __attribute__((ext_vector_type(2))) uint32_t; };
template <> struct RawT<16> { using T = __attribute__((ext_vector_type(4))) uint32_t; };
template <bool NT, class V> __device__ __forceinline__ V ld_stream(const V *p)
{
    using R = typename RawT<sizeof(V)>::T;
    if (NT) return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const R *>(p)));
    return *p;
}
template <bool NT, class V> __device__ __forceinline__ void st_stream(V *p, const V &x)
{
    using R = typename RawT<sizeof(V)>::T;
    if (NT) __builtin_nontemporal_store(__builtin_bit_cast(R, x), reinterpret_cast<R *>(p));
    else *p = x;
}

template <int VEC, int U, bool NT = false>
__device__ __forceinline__ void load_flags(FlagBlock<VEC, U> &b, const GaeArgs &a, size_t col, int t_top)
{
    using F = typename VecT<VEC>::F;
    using I = typename VecT<VEC>::I;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t o = (size_t)(t_top - u) * (size_t)a.N + col;
        if constexpr (NT) {
            b.r[u] = ld_stream<true>(reinterpret_cast<const F *>(a.r + o));
            b.v[u] = ld_stream<true>(reinterpret_cast<const F *>(a.v + o));
            b.d[u] = ld_stream<true>(reinterpret_cast<const I *>(a.done + o));
            b.tr[u] = ld_stream<true>(reinterpret_cast<const I *>(a.trunc + o));
        } else {
            b.r[u] = *reinterpret_cast<const F *>(a.r + o);
            b.v[u] = *reinterpret_cast<const F *>(a.v + o);
            b.d[u] = *reinterpret_cast<const I *>(a.done + o);
            b.tr[u] = *reinterpret_cast<const I *>(a.trunc + o);
        }
    }
}

template <int VEC, int U>
__device__ __forceinline__ void load_ev_sparse(EvBlock<VEC, U> &e, const FlagBlock<VEC, U> &b, const GaeArgs &a,
                                               size_t col, int t_top)
{
    using F = typename VecT<VEC>::F;
    int any = 0;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < VEC; ++k) any |= iget<VEC>(b.tr[u], k);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < VEC; ++k) fset<VEC>(e.ev[u], k, 0.0f);
    if (any) {  // one branch per block; only lanes that saw a truncation issue loads
#pragma unroll
        for (int u = 0; u < U; ++u)
            e.ev[u] = *reinterpret_cast<const F *>(a.ev + (size_t)(t_top - u) * (size_t)a.N + col);
    }
}

template <int VEC, int U, bool NT = false>
__device__ __forceinline__ void consume_sparse(const FlagBlock<VEC, U> &b, const EvBlock<VEC, U> &e, const GaeArgs &a,
                                               size_t col, int t_top, float (&nv)[VEC], float (&na)[VEC])
{
    using F = typename VecT<VEC>::F;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        F out, rout;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int d = iget<VEC>(b.d[u], k), tr = iget<VEC>(b.tr[u], k);
            const float ev = tr ? fget<VEC>(e.ev[u], k) : 0.0f;
            const float v = fget<VEC>(b.v[u], k);
            const float adv = gae_step(fget<VEC>(b.r[u], k), v, d, tr, ev, a.g, a.gl, nv[k], na[k]);
            fset<VEC>(out, k, adv);
            fset<VEC>(rout, k, adv + v);
        }
        const size_t o = (size_t)(t_top - u) * (size_t)a.N + col;
        // non-temporal stores in every form (round 3: 33.3 -> 32.3 us cold at [128, 65536], 127 -> 118 us at
        // [128, 262144]); NT selects non-temporal LOADS on top, which never paid
        st_stream<true>(reinterpret_cast<F *>(a.adv + o), out);
        if (a.ret) st_stream<true>(reinterpret_cast<F *>(a.ret + o), rout);
    }
}

template <int VEC, int U, bool NT = false>
__global__ void __launch_bounds__(256) k_gae_sparse(GaeArgs a)
{
    const size_t col = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (col >= (size_t)a.N) return;
    float nv[VEC], na[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { nv[k] = 0.0f; na[k] = 0.0f; }

    const int nblk = a.T / U;
    int t = a.T - 1;
    FlagBlock<VEC, U> f0, f1, f2;   // blocks b, b+1, b+2
    EvBlock<VEC, U> e0, e1;         // end values of blocks b, b+1
    if (nblk > 0) load_flags<VEC, U, NT>(f0, a, col, t);
    if (nblk > 1) load_flags<VEC, U, NT>(f1, a, col, t - U);
    if (nblk > 0) load_ev_sparse<VEC, U>(e0, f0, a, col, t);
    for (int b = 0; b < nblk; ++b) {
        if (b + 2 < nblk) load_flags<VEC, U, NT>(f2, a, col, t - 2 * U);
        if (b + 1 < nblk) load_ev_sparse<VEC, U>(e1, f1, a, col, t - U);
        consume_sparse<VEC, U, NT>(f0, e0, a, col, t, nv, na);
        f0 = f1; f1 = f2; e0 = e1;
        t -= U;
    }
    for (; t >= 0; --t) {
        FlagBlock<VEC, 1> f;
        EvBlock<VEC, 1> e;
        load_flags<VEC, 1, NT>(f, a, col, t);
        load_ev_sparse<VEC, 1>(e, f, a, col, t);
        consume_sparse<VEC, 1, NT>(f, e, a, col, t, nv, na);
    }
}

template <int VEC, int U, bool NT = false> int launch_gae_sparse(const GaeArgs &a, hipStream_t s)
{
    const long threads = a.N / VEC;
    const long waves = (threads + 63) / 64;
    const int block = waves <= 2048 ? 64 : 256;
    const long grid = (threads + block - 1) / block;
    PPO_LAUNCH((k_gae_sparse<VEC, U, NT>), dim3((unsigned)grid), dim3(block), s, a);
    return ppo_launch_status();
}

// ---------------------------------------------------------------------------------------
// Ragged form (ppo_gae_ragged: the reference's own signature, end_values[E] indexed through a running
// per-column index, src/ppo.py:468-480) on the three-stage pipeline of k_gae_sparse.  The index a row
// gathers with depends only on the FLAGS of the rows after it in time, so the gathers of block b + 1 are
// issued, per lane and only where trunc != 0, as soon as that block's flags have landed -- one block ahead
// of their use, beside the loads of block b + 2 -- instead of inside the dependent chain (round 1-2:
// k_gae<1, 8, false>, a load in the middle of every truncated step, which on this ISA also waits for every
// prefetch issued before it: 43 us at C2 against 32 for the dense form).  Same index arithmetic, same
// wrap-around of negative indices, same error flag.
template <int U> struct RagEv {
    float ev[U];
    int bad;              // a truncated row's index fell outside [-E, E): torch would raise IndexError
};

template <int U>
__device__ __forceinline__ long load_ev_ragged(RagEv<U> &e, const FlagBlock<1, U> &b, const GaeArgs &a, long idx)
{
    int any = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) any |= b.tr[u];
    e.bad = 0;
    if (any) {            // one branch per block; only lanes that saw a truncation walk their rows
#pragma unroll
        for (int u = 0; u < U; ++u) {
            e.ev[u] = 0.0f;
            if (b.tr[u]) {
                const long j = idx < 0 ? idx + a.E : idx;            // negative indices wrap (:478)
                if (j >= 0 && j < a.E) e.ev[u] = a.ev[j];
                else e.bad = 1;
            }
            idx -= (long)(b.d[u] + b.tr[u]);                          // :480
        }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            e.ev[u] = 0.0f;
            idx -= (long)b.d[u];
        }
    }
    return idx;
}

template <int U>
__device__ __forceinline__ void consume_ragged(const FlagBlock<1, U> &b, const RagEv<U> &e, const GaeArgs &a,
                                               size_t col, int t_top, float &nv, float &na)
{
    if (e.bad && a.err) *a.err = 1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float ev = b.tr[u] ? e.ev[u] : 0.0f;
        const float adv = gae_step(b.r[u], b.v[u], b.d[u], b.tr[u], ev, a.g, a.gl, nv, na);
        const size_t o = (size_t)(t_top - u) * (size_t)a.N + col;
        st_stream<true>(a.adv + o, adv);
        if (a.ret) st_stream<true>(a.ret + o, adv + b.v[u]);
    }
}

template <int U>
__global__ void __launch_bounds__(256) k_gae_ragged(GaeArgs a)
{
    const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= (size_t)a.N) return;
    float nv = 0.0f, na = 0.0f;                               // :470-471
    long idx = a.idx0[col];                                   // :469
    const int nblk = a.T / U;
    int t = a.T - 1;
    FlagBlock<1, U> f0, f1, f2;   // blocks b, b+1, b+2
    RagEv<U> e0, e1;              // end values of blocks b, b+1
    long idx_next = idx;          // index at the top row of block b+1
    if (nblk > 0) load_flags<1, U>(f0, a, col, t);
    if (nblk > 1) load_flags<1, U>(f1, a, col, t - U);
    if (nblk > 0) idx_next = load_ev_ragged<U>(e0, f0, a, idx);
    for (int b = 0; b < nblk; ++b) {
        if (b + 2 < nblk) load_flags<1, U>(f2, a, col, t - 2 * U);
        long idx_after = idx_next;
        if (b + 1 < nblk) idx_after = load_ev_ragged<U>(e1, f1, a, idx_next);
        consume_ragged<U>(f0, e0, a, col, t, nv, na);
        f0 = f1; f1 = f2; e0 = e1;
        idx_next = idx_after;
        t -= U;
    }
    for (; t >= 0; --t) {                                     // T % U leftover rows
        FlagBlock<1, 1> f;
        RagEv<1> e;
        load_flags<1, 1>(f, a, col, t);
        idx_next = load_ev_ragged<1>(e, f, a, idx_next);
        consume_ragged<1>(f, e, a, col, t, nv, na);
    }
}

template <int U> int launch_gae_ragged(const GaeArgs &a, hipStream_t s)
{
    const long waves = ((long)a.N + 63) / 64;
    const int block = waves <= 2048 ? 64 : 256;  // few columns: spread the waves over all CUs
    const long grid = ((long)a.N + block - 1) / block;
    PPO_LAUNCH((k_gae_ragged<U>), dim3((unsigned)grid), dim3(block), s, a);
    return ppo_launch_status();
}

// ---------------------------------------------------------------------------------------
// Packed-flag forms.  The whole-rollout kernel also leaves the two flags of every element as TWO BITS next to the
// i32 [T,N] buffers of the API: flag_words [ceil(T/16), N] u32, bits 2k / 2k+1 of word q of column n = done / trunc
// of row 16 q + k.  The trainer's GAE reads those words instead of the two i32 arrays: r 4 + v 4 + 0.25 read, adv 4
// written = 12.25 B per element instead of 20, and two load instructions per row instead of four.  End values are
// fetched per ROW and only by the lanes that hold a truncation in that row (the forced row T-1 is one full row; a
// TimeLimit hit elsewhere costs one sector).
//
// k_gae_tsplit: what bounds the column-per-lane scan is not bytes but loads in flight -- a wave can have 63 vector
// memory instructions outstanding, a column per lane makes every one of them a 256-byte row segment, and N columns
// are N/64 waves, so the chip holds at most 63 x 4 B x N = 16 MB in flight whatever the blocking.  Here FOUR waves
// share 64 columns and each takes a quarter of the rows: the reference's recurrence (src/ppo.py:476-486) has only
// ONE value that runs through time, adv_t = delta_t + gl (keep_t adv_{t+1}); next_value is just v_{t+1}, so
// delta_t = r_t + g (keep_t v_{t+1} + trunc_t ev_t) - v_t is computed by all four waves at once, with all their
// loads in flight together, and only the two-operation chain is handed from wave to wave through LDS (four short
// phases per 128 rows).  Every operation and its rounding is the reference's; gl * (keep * na) is evaluated as
// (gl * keep) * na, which is the same f32 for keep in {1, 0, -1}: the inner product is exact either way.
constexpr int FLAG_ROWS = 16;                       // rows per flag word

template <int VEC> struct WordT;
template <> struct WordT<1> { using W = uint32_t; };
template <> struct WordT<2> { using W = __attribute__((ext_vector_type(2))) uint32_t; };
template <> struct WordT<4> { using W = __attribute__((ext_vector_type(4))) uint32_t; };
template <int VEC> __device__ __forceinline__ uint32_t wget(const typename WordT<VEC>::W &w, int k)
{
    if constexpr (VEC == 1) return w; else return w[k];
}
template <int VEC> __device__ __forceinline__ uint32_t wany(const typename WordT<VEC>::W &w)
{
    uint32_t x = 0;
#pragma unroll
    for (int k = 0; k < VEC; ++k) x |= wget<VEC>(w, k);
    return x;
}

// rows 16 q .. 16 q + 15 of VEC columns
template <int VEC> struct WordBlock {
    typename VecT<VEC>::F r[FLAG_ROWS], v[FLAG_ROWS];
    typename WordT<VEC>::W f;
};
template <int VEC> struct EvRows {
    typename VecT<VEC>::F ev[FLAG_ROWS];
};

template <int VEC, bool FULL>
__device__ __forceinline__ void load_word_block(WordBlock<VEC> &b, const GaeArgs &a, size_t col, int q)
{
    using F = typename VecT<VEC>::F;
    b.f = *reinterpret_cast<const typename WordT<VEC>::W *>(a.flag_words + (size_t)q * (size_t)a.N + col);
#pragma unroll
    for (int u = FLAG_ROWS - 1; u >= 0; --u) {
        const int t = q * FLAG_ROWS + u;
        if (FULL || t < a.T) {
            const size_t o = (size_t)t * (size_t)a.N + col;
            b.r[u] = *reinterpret_cast<const F *>(a.r + o);
            b.v[u] = *reinterpret_cast<const F *>(a.v + o);
        }
    }
}

template <int VEC>
__device__ __forceinline__ void load_ev_word(EvRows<VEC> &e, const WordBlock<VEC> &b, const GaeArgs &a, size_t col, int q)
{
    using F = typename VecT<VEC>::F;
#pragma unroll
    for (int u = 0; u < FLAG_ROWS; ++u)
#pragma unroll
        for (int k = 0; k < VEC; ++k) fset<VEC>(e.ev[u], k, 0.0f);
    const uint32_t tr_bits = wany<VEC>(b.f) & 0xAAAAAAAAu;
    if (tr_bits) {                                  // rare below the forced last row: one test per 16 rows
#pragma unroll
        for (int u = 0; u < FLAG_ROWS; ++u)         // (a set trunc bit always belongs to a row < T)
            if (tr_bits & (2u << (2 * u)))
                e.ev[u] = *reinterpret_cast<const F *>(a.ev + (size_t)(q * FLAG_ROWS + u) * (size_t)a.N + col);
    }
}

template <int VEC, bool FULL>
__device__ __forceinline__ void consume_word_block(const WordBlock<VEC> &b, const EvRows<VEC> &e, const GaeArgs &a,
                                                   size_t col, int q, float (&nv)[VEC], float (&na)[VEC])
{
    using F = typename VecT<VEC>::F;
#pragma unroll
    for (int u = FLAG_ROWS - 1; u >= 0; --u) {
        const int t = q * FLAG_ROWS + u;
        if (!FULL && t >= a.T) continue;
        F out, rout;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const uint32_t w = wget<VEC>(b.f, k) >> (2 * u);
            const int d = (int)(w & 1u), tr = (int)((w >> 1) & 1u);
            const float ev = tr ? fget<VEC>(e.ev[u], k) : 0.0f;
            const float v = fget<VEC>(b.v[u], k);
            const float adv = gae_step(fget<VEC>(b.r[u], k), v, d, tr, ev, a.g, a.gl, nv[k], na[k]);
            fset<VEC>(out, k, adv);
            fset<VEC>(rout, k, adv + v);
        }
        const size_t o = (size_t)t * (size_t)a.N + col;
        // written once, read next by the update phase: non-temporal (A/B at C2 inside the bench step, three rounds:
        // 21.9 -> 20.5 us per launch; non-temporal LOADS of r / v on top of it: 21.5 us -- they come from the
        // Infinity Cache, where the rollout kernel has just left them)
        st_stream<true>(reinterpret_cast<F *>(a.adv + o), out);
        if (a.ret) st_stream<true>(reinterpret_cast<F *>(a.ret + o), rout);
    }
}

// Column per lane, one flag word (16 rows) per block, STAGES blocks in flight: the form for narrow batches
// (too few columns to give four waves per 64 columns something to overlap with) and the tuning baseline.
#ifdef PPO_GAE_TIMELINE                            // development (tools/gae_timeline.py): the launch's ramp and drain per wave
__device__ unsigned long long g_gae_timeline[8192 * 4];
#define TLG_MARK(k) do { if ((threadIdx.x & 63) == 0) { const size_t w_ = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; \
    if (w_ < 8192) g_gae_timeline[w_ * 4 + (k)] = wall_clock64(); } } while (0)
#else
#define TLG_MARK(k) do { } while (0)
#endif
template <int VEC, int STAGES, bool FULL>
__global__ void __launch_bounds__(256) k_gae_words(GaeArgs a)
{
    static_assert(STAGES == 2 || STAGES == 3, "");
    TLG_MARK(0);
    const size_t col = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (col >= (size_t)a.N) return;
    float nv[VEC], na[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { nv[k] = 0.0f; na[k] = 0.0f; }
    int q = (a.T + FLAG_ROWS - 1) / FLAG_ROWS - 1;
    WordBlock<VEC> f0, f1, f2;
    EvRows<VEC> e0, e1;
    load_word_block<VEC, FULL>(f0, a, col, q);
    if (q >= 1) load_word_block<VEC, FULL>(f1, a, col, q - 1);
    load_ev_word<VEC>(e0, f0, a, col, q);
    for (; q >= 0; --q) {
        if (STAGES == 3 && q >= 2) load_word_block<VEC, true>(f2, a, col, q - 2);
        if (q >= 1) load_ev_word<VEC>(e1, f1, a, col, q - 1);
        consume_word_block<VEC, FULL>(f0, e0, a, col, q, nv, na);
#ifdef PPO_GAE_TIMELINE
        if (q == (a.T + FLAG_ROWS - 1) / FLAG_ROWS - 1) TLG_MARK(1);
#endif
        f0 = f1; e0 = e1;
        if (STAGES == 3) f1 = f2;
        else if (q >= 2) load_word_block<VEC, true>(f1, a, col, q - 2);
    }
    TLG_MARK(2);
#ifdef PPO_GAE_TIMELINE
    __builtin_amdgcn_s_waitcnt(0x0F70);            // every store of the wave has landed
    TLG_MARK(3);
#endif
}

// Buffer addressing for the time-split kernel: one resource descriptor per array (4 SGPRs), the row's byte offset in
// an SGPR, the lane's byte offset inside the row in ONE VGPR -- instead of a 64-bit VGPR pair per row in flight.
// Raw buffer, stride 0, num_records = the array's size in bytes: an access beyond it reads 0 / is dropped.
using BufRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ BufRsrc make_rsrc(const void *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
template <int VEC> __device__ __forceinline__ typename RawT<4 * VEC>::T buf_ld_raw(BufRsrc rs, uint32_t voff, uint32_t soff)
{
    if constexpr (VEC == 1) return __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0);
    else if constexpr (VEC == 2) return __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    else return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
}
template <int VEC> __device__ __forceinline__ typename VecT<VEC>::F buf_ld(BufRsrc rs, uint32_t voff, uint32_t soff)
{
    return __builtin_bit_cast(typename VecT<VEC>::F, buf_ld_raw<VEC>(rs, voff, soff));
}
template <int VEC> __device__ __forceinline__ void buf_st(BufRsrc rs, uint32_t voff, uint32_t soff, const typename VecT<VEC>::F &x)
{
    const auto raw = __builtin_bit_cast(typename RawT<4 * VEC>::T, x);
    constexpr int AUX = 2;                         // nt: advantages / returns are written once and read much later
    if constexpr (VEC == 1) __builtin_amdgcn_raw_buffer_store_b32(raw, rs, voff, soff, AUX);
    else if constexpr (VEC == 2) __builtin_amdgcn_raw_buffer_store_b64(raw, rs, voff, soff, AUX);
    else __builtin_amdgcn_raw_buffer_store_b128(raw, rs, voff, soff, AUX);
}

// Four waves per 64 * VEC columns, WPW flag words (16 * WPW rows) per wave and pass; see the head of this section.
template <int VEC, int WPW, bool FULL>
__global__ void __launch_bounds__(256) k_gae_tsplit(GaeArgs a)
{
    using F = typename VecT<VEC>::F;
    using W = typename WordT<VEC>::W;
    constexpr int R = FLAG_ROWS * WPW;
    __shared__ float carry[VEC][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t col = ((size_t)blockIdx.x * 64 + lane) * VEC;
    const bool live = col < (size_t)a.N;            // (no early return: every wave takes part in the barriers)
    const uint32_t off = (uint32_t)col * 4u;        // byte offset inside a row (the host checks N * 4 < 2^32)
    const uint32_t stride = (uint32_t)a.N * 4u;     // bytes per row (the host checks T * N * 4 < 2^32)
    const int words = (a.T + FLAG_ROWS - 1) / FLAG_ROWS;
    const uint32_t bytes = (uint32_t)a.T * stride;
    const BufRsrc rs_r = make_rsrc(a.r, bytes), rs_v = make_rsrc(a.v, bytes), rs_ev = make_rsrc(a.ev, bytes);
    const BufRsrc rs_f = make_rsrc(a.flag_words, (uint32_t)words * stride);
    const BufRsrc rs_adv = make_rsrc(a.adv, bytes), rs_ret = make_rsrc(a.ret, a.ret ? bytes : 0u);
    auto row = [&](int t) { return (uint32_t)t * stride; };
    bool first = true;                              // nothing has been scanned yet: advantages start from 0 (:471)
    for (int qtop = words - 1; qtop >= 0; qtop -= 4 * WPW) {
        const int q_hi = qtop - w * WPW;            // this wave's words: q_hi, q_hi - 1, .. (those >= 0)
        F r[R], v[R], ev[R];
        W f[WPW];
        F v_above;                                  // values of the row above this wave's top row: its next_value (:485)
#pragma unroll
        for (int k = 0; k < VEC; ++k) fset<VEC>(v_above, k, 0.0f);
        if (live && q_hi >= 0) {
            const int t_above = (q_hi + 1) * FLAG_ROWS;
            if (t_above < a.T) v_above = buf_ld<VEC>(rs_v, off, row(t_above));
#pragma unroll
            for (int j = 0; j < WPW; ++j) {
                const int q = q_hi - j;
                if (q < 0) continue;
                f[j] = __builtin_bit_cast(W, buf_ld_raw<VEC>(rs_f, off, row(q)));
#pragma unroll
                for (int u = FLAG_ROWS - 1; u >= 0; --u) {
                    const int t = q * FLAG_ROWS + u, i = j * FLAG_ROWS + (FLAG_ROWS - 1 - u);   // i: top row first
                    if (FULL || t < a.T) {
                        r[i] = buf_ld<VEC>(rs_r, off, row(t));
                        v[i] = buf_ld<VEC>(rs_v, off, row(t));
                    }
                }
            }
            // end values of this wave's truncated cells
#pragma unroll
            for (int j = 0; j < WPW; ++j) {
                const int q = q_hi - j;
                if (q < 0) continue;
                const uint32_t tr_bits = wany<VEC>(f[j]) & 0xAAAAAAAAu;
#pragma unroll
                for (int u = FLAG_ROWS - 1; u >= 0; --u) {
                    const int i = j * FLAG_ROWS + (FLAG_ROWS - 1 - u);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) fset<VEC>(ev[i], k, 0.0f);
                }
                if (tr_bits) {
#pragma unroll
                    for (int u = FLAG_ROWS - 1; u >= 0; --u)
                        if (tr_bits & (2u << (2 * u)))
                            ev[j * FLAG_ROWS + (FLAG_ROWS - 1 - u)] = buf_ld<VEC>(rs_ev, off, row(q * FLAG_ROWS + u));
                }
            }
            // delta_t of every row (:476-482), all four waves at once; r[i] <- delta
#pragma unroll
            for (int j = 0; j < WPW; ++j) {
                const int q = q_hi - j;
                if (q < 0) continue;
#pragma unroll
                for (int u = FLAG_ROWS - 1; u >= 0; --u) {
                    const int t = q * FLAG_ROWS + u, i = j * FLAG_ROWS + (FLAG_ROWS - 1 - u);
                    if (!FULL && t >= a.T) continue;
                    F delta;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        const uint32_t fw = wget<VEC>(f[j], k) >> (2 * u);
                        const int d = (int)(fw & 1u), tr = (int)((fw >> 1) & 1u);
                        const float keep = (float)(1 - (d + tr));                               // :476-477
                        // the row above: v[i - 1], or v_above for the wave's top row (rows >= T do not exist: the
                        // top row of the rollout sees next_value = 0, :470)
                        float nv;
                        if (i == 0) nv = fget<VEC>(v_above, k);
                        else nv = (FULL || t + 1 < a.T) ? fget<VEC>(v[i - 1], k) : 0.0f;
                        const float e = tr ? fget<VEC>(ev[i], k) : 0.0f;
                        nv = keep * nv + (float)tr * e;                                         // :478
                        fset<VEC>(delta, k, fget<VEC>(r[i], k) + a.g * nv - fget<VEC>(v[i], k));   // :482
                    }
                    r[i] = delta;
                }
            }
        }
        // the chain through time, a wave at a time, latest rows first
#pragma unroll 1
        for (int p = 0; p < 4; ++p) {
            if (p == w && live && q_hi >= 0) {
                float na[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) na[k] = first && w == 0 ? 0.0f : carry[k][lane];
#pragma unroll
                for (int j = 0; j < WPW; ++j) {
                    const int q = q_hi - j;
                    if (q < 0) continue;
#pragma unroll
                    for (int u = FLAG_ROWS - 1; u >= 0; --u) {
                        const int t = q * FLAG_ROWS + u, i = j * FLAG_ROWS + (FLAG_ROWS - 1 - u);
                        if (!FULL && t >= a.T) continue;
                        F out, rout;
#pragma unroll
                        for (int k = 0; k < VEC; ++k) {
                            const uint32_t fw = wget<VEC>(f[j], k) >> (2 * u);
                            const float keep = (float)(1 - (int)((fw & 1u) + ((fw >> 1) & 1u)));
                            const float c = a.gl * keep;                                        // exact: keep in {1, 0, -1}
                            const float adv = fget<VEC>(r[i], k) + c * na[k];                   // :479, :483
                            na[k] = adv;                                                        // :486
                            fset<VEC>(out, k, adv);
                            fset<VEC>(rout, k, adv + fget<VEC>(v[i], k));
                        }
                        buf_st<VEC>(rs_adv, off, row(t), out);
                        if (a.ret) buf_st<VEC>(rs_ret, off, row(t), rout);
                    }
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) carry[k][lane] = na[k];
            }
            __syncthreads();
        }
        first = false;
    }
}

template <int VEC, int STAGES> int launch_gae_words(const GaeArgs &a, hipStream_t s)
{
    const long threads = a.N / VEC;
    const int block = (threads + 63) / 64 <= 2048 ? 64 : 256;
    const long grid = (threads + block - 1) / block;
    if (a.T % FLAG_ROWS == 0) PPO_LAUNCH((k_gae_words<VEC, STAGES, true>), dim3((unsigned)grid), dim3(block), s, a);
    else PPO_LAUNCH((k_gae_words<VEC, STAGES, false>), dim3((unsigned)grid), dim3(block), s, a);
    return ppo_launch_status();
}

template <int VEC, int WPW> int launch_gae_tsplit(const GaeArgs &a, hipStream_t s)
{
    if ((size_t)a.T * (size_t)a.N * 4 >= ((size_t)1 << 32)) return launch_gae_words<VEC, 2>(a, s);   // 32-bit row offsets
    const long grid = (a.N / VEC + 63) / 64;
    if (a.T % FLAG_ROWS == 0) PPO_LAUNCH((k_gae_tsplit<VEC, WPW, true>), dim3((unsigned)grid), dim3(256), s, a);
    else PPO_LAUNCH((k_gae_tsplit<VEC, WPW, false>), dim3((unsigned)grid), dim3(256), s, a);
    return ppo_launch_status();
}

// Variant table of the packed forms (tools/tune_gae_packed.py; numbers in DESIGN.md)
int launch_gae_packed_variant(const GaeArgs &a, int variant, hipStream_t s)
{
    switch (variant) {
    case 0: return launch_gae_words<1, 3>(a, s);
    case 1: return launch_gae_words<2, 2>(a, s);
    case 2: return launch_gae_words<1, 2>(a, s);
    case 3: return launch_gae_tsplit<1, 2>(a, s);
    case 4: return launch_gae_tsplit<1, 1>(a, s);
    case 5: return launch_gae_tsplit<2, 1>(a, s);
    default: return (int)hipErrorInvalidValue;
    }
}

int packed_variant_vec(int variant)
{
    switch (variant) {
    case 1: case 5: return 2;
    default: return 1;
    }
}

// A pure streaming kernel with the GAE traffic mix (5 x 4 B read, 4 B written per element,
// 16 B per lane, full occupancy): the measured ceiling the scan kernels are compared to.
__global__ void __launch_bounds__(256) k_stream_ceiling(const float4 *r, const float4 *v, const float4 *ev,
                                                        const int4 *d, const int4 *tr, float4 *out, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = r[i], b = v[i], c = ev[i];
        const int4 e = d[i], f = tr[i];
        out[i] = make_float4(a.x + b.x + c.x + (float)(e.x + f.x), a.y + b.y + c.y + (float)(e.y + f.y),
                             a.z + b.z + c.z + (float)(e.z + f.z), a.w + b.w + c.w + (float)(e.w + f.w));
    }
}

// --- ragged pre-pass: ep_counts (src/ppo.py:468) and cumsum - 1 (:469) ------------
// k_ep_counts: one lane per column, eight rows of flags in flight; also the total of each block of
// 256 columns.  k_scan_apply: every block adds up the totals of the blocks before it and scans its
// own 256 counts (shuffles + one LDS stage) -- no serial single-workgroup pass.
constexpr int RAG_BLOCK = 256;

__device__ __forceinline__ long block_sum_256(long x, long *lds4)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, PPO_WAVE);
    __syncthreads();                       // lds4 may still be read from a previous use
    if (lane == 0) lds4[wid] = x;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

__global__ void __launch_bounds__(RAG_BLOCK) k_ep_counts(const int *done, const int *trunc, long *counts, long *totals,
                                                         int T, int N)
{
    __shared__ long lds4[4];
    const size_t n = (size_t)blockIdx.x * RAG_BLOCK + threadIdx.x;
    long c = 0;
    if (n < (size_t)N) {
        int t = 0;
        for (; t + 8 <= T; t += 8) {
            int d[8], tr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t o = (size_t)(t + u) * N + n;
                d[u] = done[o];
                tr[u] = trunc[o];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) c += d[u] + tr[u] - d[u] * tr[u];
        }
        for (; t < T; ++t) {
            const size_t o = (size_t)t * N + n;
            const int d = done[o], tr = trunc[o];
            c += d + tr - d * tr;
        }
        counts[n] = c;
    }
    const long tot = block_sum_256(c, lds4);
    if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}

// In place: counts[n] <- cumsum(counts)[n] - 1.
__global__ void __launch_bounds__(RAG_BLOCK) k_scan_apply(long *counts, const long *totals, int N)
{
    __shared__ long lds4[4];
    __shared__ long wave_tot[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    long before = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += RAG_BLOCK) before += totals[b];
    before = block_sum_256(before, lds4);
    const size_t n = (size_t)blockIdx.x * RAG_BLOCK + threadIdx.x;
    const long c = n < (size_t)N ? counts[n] : 0;
    long incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const long y = __shfl_up(incl, off, PPO_WAVE);
        if (lane >= off) incl += y;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    long wave_off = 0;
    for (int w = 0; w < wid; ++w) wave_off += wave_tot[w];
    if (n < (size_t)N) counts[n] = before + wave_off + incl - 1;
}

template <int VEC, int U, bool DENSE_EV> int launch_gae(const GaeArgs &a, hipStream_t s)
{
    const long threads = a.N / VEC;
    const long waves = (threads + 63) / 64;
    const int block = waves <= 2048 ? 64 : 256;  // few columns: spread the waves over all CUs
    const long grid = (threads + block - 1) / block;
    PPO_LAUNCH((k_gae<VEC, U, DENSE_EV>), dim3((unsigned)grid), dim3(block), s, a);
    return ppo_launch_status();
}

// Variant table (tools/tune_gae.py; numbers in DESIGN.md).  0-1 stream the dense end-value
// buffer, 2-7 fetch it only where a truncation is flagged; 6-7 use non-temporal accesses (round 2 also measured
// four-stage, 16-row, four-column and forced-workgroup-size forms, numbered 8-14 in profiles/r02_gae_variants_*.txt:
// none faster, removed) (measured
// inside bench.py: <2,8> 34.0 us, <2,8,nt> 39.5, <1,8> 35.2, <1,8,nt> 33.5 -- no clear win, and the
// <2,8> form is sensitive to how the compiler schedules its loads, so its source is left as it is).
int launch_gae_dense_variant(const GaeArgs &a, int variant, hipStream_t s)
{
    switch (variant) {
    case 0: return launch_gae<1, 8, true>(a, s);
    case 1: return launch_gae<2, 4, true>(a, s);
    case 2: return launch_gae_sparse<1, 4>(a, s);
    case 3: return launch_gae_sparse<1, 8>(a, s);
    case 4: return launch_gae_sparse<2, 4>(a, s);
    case 5: return launch_gae_sparse<2, 8>(a, s);
    case 6: return launch_gae_sparse<2, 8, true>(a, s);
    case 7: return launch_gae_sparse<1, 8, true>(a, s);
    default: return (int)hipErrorInvalidValue;
    }
}

int variant_vec(int variant)
{
    return (variant == 1 || variant == 4 || variant == 5 || variant == 6) ? 2 : 1;
}

}  // namespace

// ---- allocation-free form of ppo_gae_ragged -----------------------------------------------------
// The ragged scan's only scratch is idx0[N] + one total per 256 columns.  Here it lives in a
// file-scope device array (one copy per GPU, held until the process ends) instead of a caller
// workspace, for callers that cannot allocate.  The array's NAME is only a host-side handle: the
// device address comes from hipGetSymbolAddress, once per device, and is what the kernels receive.
constexpr int RAGGED_NOALLOC_MAX_N = PPO_GAE_RAGGED_NOALLOC_MAX_N;   // 262144: BASELINE configs[2]'s whole batch; 2.1 MB
constexpr size_t RAGGED_NOALLOC_BYTES =
    ((size_t)RAGGED_NOALLOC_MAX_N + (RAGGED_NOALLOC_MAX_N + RAG_BLOCK - 1) / RAG_BLOCK) * sizeof(long);
static __device__ __align__(256) unsigned char scratch[RAGGED_NOALLOC_BYTES];
constexpr int NOALLOC_MAX_DEVICES = 64;
static void *g_scratch_ptr[NOALLOC_MAX_DEVICES];  // per device: address of that GPU's copy of `scratch`

static int ragged_scratch(void **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) return (int)hipErrorInvalidDevice;
    if (!g_scratch_ptr[dev]) {                     // first call on this device, with it current
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!p) return (int)hipErrorInvalidSymbol;
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" {

static int gae_dense_launch(const float *rewards, const float *values, const float *end_value_dense,
                            const int *done, const int *trunc, double gamma, double lam, float *adv,
                            float *ret, int T, int N, int variant, void *stream)
{
    PPO_REQUIRE(rewards && values && end_value_dense && done && trunc && adv);
    PPO_REQUIRE(T >= 0 && N >= 0);
    if (T == 0 || N == 0) return 0;
    const int vec = variant_vec(variant);
    const size_t al = 4 * (size_t)vec;
    PPO_REQUIRE(N % vec == 0);
    PPO_REQUIRE(ppo_aligned(rewards, al) && ppo_aligned(values, al) && ppo_aligned(end_value_dense, al) &&
                ppo_aligned(done, al) && ppo_aligned(trunc, al) && ppo_aligned(adv, al) &&
                (!ret || ppo_aligned(ret, al)));
    GaeArgs a{rewards, values, end_value_dense, done, trunc, nullptr, adv, ret, nullptr, 0, nullptr,
              (float)gamma, (float)(gamma * lam), T, N};
    return launch_gae_dense_variant(a, variant, (hipStream_t)stream);
}

// Measured on MI355X at T=128, N=65536: sparse end values, two columns per lane,
// 8-row blocks is the fastest form; it needs an even N and 8-byte aligned bases.
static int gae_default_variant(const float *rewards, const float *values, const float *end_value_dense,
                               const int *done, const int *trunc, const float *adv, const float *ret, int N)
{
    const bool vec2 = (N % 2 == 0) && ppo_aligned(rewards, 8) && ppo_aligned(values, 8) &&
                      ppo_aligned(end_value_dense, 8) && ppo_aligned(done, 8) && ppo_aligned(trunc, 8) &&
                      ppo_aligned(adv, 8) && (!ret || ppo_aligned(ret, 8));
    return vec2 ? 5 : 3;
}

// Tuning hook (not part of the public header): pick a (VEC, U) instantiation explicitly.
int ppo_gae_dense_variant(const float *rewards, const float *values, const float *end_value_dense,
                          const int *done, const int *trunc, double gamma, double lam, float *adv,
                          float *ret, int T, int N, int variant, void *stream)
{
    return gae_dense_launch(rewards, values, end_value_dense, done, trunc, gamma, lam, adv, ret, T, N, variant, stream);
}

int ppo_gae_dense(const float *rewards, const float *values, const float *end_value_dense,
                  const int *done, const int *trunc, double gamma, double lam, float *adv,
                  float *ret, int T, int N, void *stream)
{
    PPO_REQUIRE(rewards && values && end_value_dense && done && trunc && adv);
    return gae_dense_launch(rewards, values, end_value_dense, done, trunc, gamma, lam, adv, ret, T, N,
                            gae_default_variant(rewards, values, end_value_dense, done, trunc, adv, ret, N), stream);
}

// ---- packed-flag forms (see k_gae_tsplit) ------------------------------------------------------
static int gae_packed_launch(const float *rewards, const float *values, const float *end_value_dense,
                             const uint32_t *flag_words, double gamma, double lam, float *adv, float *ret, int T, int N,
                             int variant, void *stream)
{
    PPO_REQUIRE(rewards && values && end_value_dense && flag_words && adv);
    PPO_REQUIRE(T >= 0 && N >= 0);
    if (T == 0 || N == 0) return 0;
    const int vec = packed_variant_vec(variant);
    const size_t al = 4 * (size_t)vec;
    PPO_REQUIRE(N % vec == 0 && (size_t)N * 4 < ((size_t)1 << 32));
    PPO_REQUIRE(ppo_aligned(rewards, al) && ppo_aligned(values, al) && ppo_aligned(end_value_dense, al) &&
                ppo_aligned(flag_words, al) && ppo_aligned(adv, al) && (!ret || ppo_aligned(ret, al)));
    GaeArgs a{rewards, values, end_value_dense, nullptr, nullptr, flag_words, adv, ret, nullptr, 0, nullptr,
              (float)gamma, (float)(gamma * lam), T, N};
    return launch_gae_packed_variant(a, variant, (hipStream_t)stream);
}

// Tuning hook (not part of the public header)
int ppo_gae_dense_packed_variant(const float *rewards, const float *values, const float *end_value_dense,
                                 const uint32_t *flag_words, double gamma, double lam, float *adv, float *ret, int T,
                                 int N, int variant, void *stream)
{
    return gae_packed_launch(rewards, values, end_value_dense, flag_words, gamma, lam, adv, ret, T, N, variant, stream);
}

size_t ppo_gae_flag_words(int T, int N)
{
    return T > 0 && N > 0 ? (size_t)((T + FLAG_ROWS - 1) / FLAG_ROWS) * (size_t)N : 0;
}

int ppo_gae_dense_packed(const float *rewards, const float *values, const float *end_value_dense,
                         const uint32_t *flag_words, double gamma, double lam, float *adv, float *ret, int T, int N,
                         void *stream)
{
    PPO_REQUIRE(rewards && values && end_value_dense && flag_words && adv);
    PPO_REQUIRE(ppo_aligned(rewards, 4) && ppo_aligned(values, 4) && ppo_aligned(end_value_dense, 4) &&
                ppo_aligned(flag_words, 4) && ppo_aligned(adv, 4) && (!ret || ppo_aligned(ret, 4)));
    // Measured (profiles/r03_gae_packed_variants.txt).  Narrow batches are too few waves for a column per lane to keep
    // the memory system busy: the time-split kernel gives them four waves per 64 columns (13.0 vs 15.1 us at
    // [128, 16384], 19.5 vs 26.6 at [256, 4096], 139 vs 198 at [2048, 1024]); at 65536 columns the column-per-lane
    // form is ahead (23.6 vs 28.8 us on HBM-cold inputs), from 131072 on with two columns per lane (77 vs 87 us at
    // [128, 262144]).
    const bool even = (N % 2 == 0) && ppo_aligned(rewards, 8) && ppo_aligned(values, 8) && ppo_aligned(end_value_dense, 8) &&
                      ppo_aligned(flag_words, 8) && ppo_aligned(adv, 8) && (!ret || ppo_aligned(ret, 8));
    const int variant = N <= 8192 ? 3 : N <= 32768 ? 4 : (N >= 131072 && even) ? 1 : 0;
    return gae_packed_launch(rewards, values, end_value_dense, flag_words, gamma, lam, adv, ret, T, N, variant, stream);
}

// Bench-only (private): the next `count` kernel launches of this library are timed into event slots
// first_slot .. first_slot + count - 1 (PPO_LAUNCH in common.h); read a slot back with ppo_prof_elapsed_us
// (which waits for it).  Single bench thread; slots are process-wide.
int ppo_prof_arm(int first_slot, int count)
{
    PPO_REQUIRE(first_slot >= 0 && count >= 0 && first_slot + count <= PPO_PROF_SLOTS);
    for (int k = first_slot; k < first_slot + count; ++k) {
        if (g_ppo_prof.have[k]) continue;
        hipError_t e = hipEventCreate(&g_ppo_prof.start[k]);
        if (e == hipSuccess) e = hipEventCreate(&g_ppo_prof.stop[k]);
        if (e != hipSuccess) return (int)e;
        g_ppo_prof.have[k] = true;
    }
    g_ppo_prof.next = first_slot;
    g_ppo_prof.left = count;
    return 0;
}

// Slot the next armed launch would use (= first_slot + launches timed since ppo_prof_arm): tells a caller that armed a
// whole step which slot its last launch landed in
int ppo_prof_next(void) { return g_ppo_prof.next; }

// ppo_gae_dense whose dispatch is timed into event slot `slot`
int ppo_gae_dense_timed(const float *rewards, const float *values, const float *end_value_dense,
                        const int *done, const int *trunc, double gamma, double lam, float *adv,
                        float *ret, int T, int N, int slot, void *stream)
{
    PPO_REQUIRE(rewards && values && end_value_dense && done && trunc && adv);
    PPO_REQUIRE(slot >= 0 && slot < PPO_PROF_SLOTS && T > 0 && N > 0);
    const int rc = ppo_prof_arm(slot, 1);
    if (rc) return rc;
    const int out = ppo_gae_dense(rewards, values, end_value_dense, done, trunc, gamma, lam, adv, ret, T, N, stream);
    g_ppo_prof.left = 0;                           // (an argument check may have returned before the launch)
    return out;
}

int ppo_prof_elapsed_us(int slot, float *us)
{
    PPO_REQUIRE(slot >= 0 && slot < PPO_PROF_SLOTS && g_ppo_prof.have[slot] && us);
    hipError_t e = hipEventSynchronize(g_ppo_prof.stop[slot]);
    if (e != hipSuccess) return (int)e;
    float ms = 0.0f;
    e = hipEventElapsedTime(&ms, g_ppo_prof.start[slot], g_ppo_prof.stop[slot]);
    *us = ms * 1000.0f;
    return (int)e;
}

// Tuning hook: streaming ceiling with the GAE traffic mix (n elements, 16-byte aligned, n % 4 == 0).
int ppo_stream_ceiling(const float *r, const float *v, const float *ev, const int *d, const int *tr, float *out,
                       long n, void *stream)
{
    PPO_REQUIRE(r && v && ev && d && tr && out && n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(k_stream_ceiling, dim3(2048), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(r), reinterpret_cast<const float4 *>(v),
                       reinterpret_cast<const float4 *>(ev), reinterpret_cast<const int4 *>(d),
                       reinterpret_cast<const int4 *>(tr), reinterpret_cast<float4 *>(out), (size_t)n / 4);
    return ppo_launch_status();
}

size_t ppo_gae_ragged_workspace_bytes(int N)
{
    return N > 0 ? ((size_t)N + (size_t)((N + RAG_BLOCK - 1) / RAG_BLOCK)) * sizeof(long) : 0;
}

int ppo_gae_ragged_noalloc(const float *rewards, const float *values, const float *end_values, long E,
                           const int *done, const int *trunc, double gamma, double lam, float *adv, int T,
                           int N, int *err_flag, void *stream)
{
    PPO_REQUIRE(N <= RAGGED_NOALLOC_MAX_N);        // beyond the scratch array: use ppo_gae_ragged with a workspace
    void *ws = nullptr;
    const int rc = ragged_scratch(&ws);
    if (rc) return rc;
    // every element of the scratch a launch reads (idx0[0..N), totals[0..ceil(N/256))) is written by
    // k_ep_counts of the same call first, so nothing has to be cleared between calls
    return ppo_gae_ragged(rewards, values, end_values, E, done, trunc, gamma, lam, adv, T, N, ws,
                          RAGGED_NOALLOC_BYTES, err_flag, stream);
}

int ppo_gae_ragged(const float *rewards, const float *values, const float *end_values, long E,
                   const int *done, const int *trunc, double gamma, double lam, float *adv, int T,
                   int N, void *workspace, size_t workspace_bytes, int *err_flag, void *stream)
{
    PPO_REQUIRE(rewards && values && done && trunc && adv);
    PPO_REQUIRE(T >= 0 && N >= 0 && E >= 0);
    PPO_REQUIRE(E == 0 || end_values);
    if (T == 0 || N == 0) return 0;
    PPO_REQUIRE(workspace && workspace_bytes >= ppo_gae_ragged_workspace_bytes(N));
    PPO_REQUIRE(ppo_aligned(workspace, 8));
    hipStream_t s = (hipStream_t)stream;
    long *idx0 = reinterpret_cast<long *>(workspace);
    long *totals = idx0 + N;
    const unsigned nb = (unsigned)((N + RAG_BLOCK - 1) / RAG_BLOCK);
    hipLaunchKernelGGL(k_ep_counts, dim3(nb), dim3(RAG_BLOCK), 0, s, done, trunc, idx0, totals, T, N);
    int rc = ppo_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(RAG_BLOCK), 0, s, idx0, totals, N);
    rc = ppo_launch_status();
    if (rc) return rc;
    GaeArgs a{rewards, values, end_values, done, trunc, nullptr, adv, nullptr, idx0, E, err_flag,
              (float)gamma, (float)(gamma * lam), T, N};
    return launch_gae_ragged<8>(a, s);
}

}  // extern "C"

#ifdef PPO_GAE_TIMELINE
extern "C" int ppo_gae_timeline_read(unsigned long long *dst, int words)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gae_timeline), (size_t)words * 8, 0, hipMemcpyDeviceToHost);
}
#endif
