// Shared device helpers for the gfx950 PPO kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/ppo_amd.h"

#define PPO_WAVE 64

#define PPO_REQUIRE(cond)                          \
    do {                                           \
        if (!(cond)) return (int)hipErrorInvalidValue; \
    } while (0)

static inline int ppo_launch_status() { return (int)hipGetLastError(); }

static inline bool ppo_aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// ---------------------------------------------------------------------------
// Bench-only kernel timing (private symbols ppo_prof_arm / ppo_prof_elapsed_us / ppo_gae_dense_timed; not part
// of include/ppo_amd.h).  ppo_prof_arm(first, count) makes the next `count` kernel launches that go through
// PPO_LAUNCH carry a start/stop event pair (hipExtLaunchKernelGGL timestamps the dispatch packet itself; a
// plain hipEventRecord pair around a launch also times two barrier packets, +25..50 us on this stack;
// rocprofv3 --kernel-trace agrees with this form).  The slots are the library's only process-wide
// state: one bench thread, not thread-safe.  Un-armed (always, outside bench.py) the macro is a plain launch.
constexpr int PPO_PROF_SLOTS = 256;
struct PpoProf {
    hipEvent_t start[PPO_PROF_SLOTS], stop[PPO_PROF_SLOTS];
    bool have[PPO_PROF_SLOTS];
    int next, left;                // slot of the next armed launch; armed launches still to come
};
extern PpoProf g_ppo_prof;         // gae.hip
static inline int ppo_prof_take()
{
    if (g_ppo_prof.left <= 0) return -1;
    g_ppo_prof.left -= 1;
    return g_ppo_prof.next++;
}
#define PPO_LAUNCH(kern, grid, block, stream, ...)                                                              \
    do {                                                                                                        \
        const int ps_ = ppo_prof_take();                                                                        \
        if (ps_ >= 0)                                                                                           \
            hipExtLaunchKernelGGL(kern, grid, block, 0, stream, g_ppo_prof.start[ps_], g_ppo_prof.stop[ps_], 0, \
                                  __VA_ARGS__);                                                                 \
        else                                                                                                    \
            hipLaunchKernelGGL(kern, grid, block, 0, stream, __VA_ARGS__);                                      \
    } while (0)

// ---------------------------------------------------------------------------
// Philox4x32-10 (counter-based; integer arithmetic only, so host oracle and device
// agree bit for bit).
struct Philox4 {
    uint32_t v[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                  uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// Two 32-bit words -> double in [0,1) with 53 random bits (numpy Generator.random()).
__device__ __forceinline__ double u53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// Four reset uniforms for (seed; env_id, episode): counter = (env_lo, env_hi, episode, block).
__device__ __forceinline__ void reset_uniform4(uint64_t seed, uint64_t env_id, uint32_t episode, double u[4])
{
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const Philox4 a = philox4x32_10((uint32_t)env_id, (uint32_t)(env_id >> 32), episode, 0u, k0, k1);
    const Philox4 b = philox4x32_10((uint32_t)env_id, (uint32_t)(env_id >> 32), episode, 1u, k0, k1);
    u[0] = u53(a.v[0], a.v[1]);
    u[1] = u53(a.v[2], a.v[3]);
    u[2] = u53(b.v[0], b.v[1]);
    u[3] = u53(b.v[2], b.v[3]);
}

// ---------------------------------------------------------------------------
// Wave64 butterfly sum (the compiler lowers the xor-shuffles to DPP / permlane ops).
__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, PPO_WAVE);
    return x;
}
