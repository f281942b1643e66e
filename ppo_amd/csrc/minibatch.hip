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
    a.state_vec4 = states && state_words == 4 && ppo_aligned(states, 16) && ppo_aligned(mb_states, 16);
    a.action_vec4 = actions && action_words == 4 && ppo_aligned(actions, 16) && ppo_aligned(mb_actions, 16);
    long blocks = (B + MB_BLOCK - 1) / MB_BLOCK;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_minibatch_gather, dim3((unsigned)blocks), dim3(MB_BLOCK), 0, (hipStream_t)stream, a);
    return ppo_launch_status();
}

}  // extern "C"
