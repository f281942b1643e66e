/*
 * ppo_amd.h -- C ABI of libppo_amd.so: the MI355X (gfx950) kernels for the PPO
 * rollout-and-advantage hot path of BradleyGS1/PPO.
 *
 * The reference is pure Python and has no FFI of its own; every entry
 * point below cites the reference code (file:line under /root/reference) whose work it
 * takes over.  The Python surface in ppo_amd/ppo.py (same class/method names as
 * src/ppo.py) binds these with ctypes (ppo_amd/_lib.py); INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer (hipMalloc / torch tensor.data_ptr()) unless
 *     marked "host".  The library never allocates, frees or retains caller memory and
 *     never synchronises: every call enqueues kernels on `stream` and returns.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - Return value: 0 = hipSuccess, otherwise a hipError_t (hipErrorInvalidValue = 1
 *     for rejected arguments; nothing has been launched in that case).
 *   - Matrices are row-major [T, N]: element (t, n) at t*N + n  (the layout of the
 *     reference's rollout buffers, src/ppo.py:309-315).
 *   - The entry points declared here keep no state between calls: what a call does depends on
 *     its arguments only (no environment variables, no mode switches).  One host thread per
 *     device (the reference is single-threaded, so no concurrency contract is inherited).
 *     Exceptions, both stated where they are declared: ppo_gae_ragged_noalloc works in a
 *     per-device scratch array inside the library, and the private bench hooks listed in
 *     INTEGRATION.md (not declared here) keep process-wide HIP event slots and are for one
 *     bench thread only.
 */
#ifndef PPO_AMD_H
#define PPO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: the sampling stream of ppo_categorical_sample / ppo_rollout_fused draws one Philox block per FOUR policy
 *    steps (counter = step / 4, word = step % 4; version 1 drew one block per step), so a run resumed across the
 *    two versions samples different actions; PPO_NET_BF16X3 added; private hooks changed.
 * 3: ppo_episode_stats_incremental_ints, ppo_episode_stats_ints_hist_offset added; nothing else changed.
 * 4: the direct histogram behind ppo_episode_stats_ints_hist_offset is kept as PPO_STATS_INTS_REPLICAS partial
 *    histograms (their u32 sum is the window's); nothing else changed. */
#define PPO_AMD_ABI_VERSION 4

#define PPO_ENV_CARTPOLE 0 /* CartPole-v1: obs_dim 4, actions {0,1}   */
#define PPO_ENV_ACROBOT 1  /* Acrobot-v1:  obs_dim 6, actions {0,1,2} */

/* How ppo_rollout_fused multiplies inside the policy / value network (the result is f32 either way) */
#define PPO_NET_F32 0        /* v_mfma_f32_32x32x2_f32: every product an exact f32 FMA */
#define PPO_NET_F16_SPLIT 1  /* v_mfma_f32_32x32x16_f16 on two-piece f16 splits of both operands
                                (22 significant bits per operand, f32 accumulate): ~1.9x faster.
                                As accurate as f32 unless the backbone weights are tiny: see
                                split_diag of ppo_rollout_pack_params */
#define PPO_NET_BF16X3 2     /* v_mfma_f32_32x32x16_bf16 on three-piece bf16 splits of both operands: every f32
                                operand is represented exactly, a product is its six largest cross terms in one
                                f32 accumulator.  As accurate as PPO_NET_F32 at every weight scale (measured),
                                ~1.7x faster than it: the f32-grade mode, and what to use outside the limits of
                                PPO_NET_F16_SPLIT */

int ppo_abi_version(void);

/* Observation width of an env kind (4 / 6), or -1. */
int ppo_env_obs_dim(int kind);

/*
 * Reset N envs and write their first observations.
 * Replaces SyncVecEnv.vec_reset (src/ppo.py:269-275): N x env.reset() + packing into
 * states[N, obs_dim] f32.
 *   state      [N,4] f64  physics state (fp64 like gymnasium's), overwritten
 *   elapsed    [N]   i32  TimeLimit step counter, zeroed
 *   ep_ret     [N]   f32, ep_len [N] i32: RecordEpisodeStatistics accumulators, zeroed
 *   episode    [N]   u32  reset counter: read as the Philox stream position, then +1
 *   reset_draw [N,4] f64  uniforms in [0,1) to build the initial state from, or NULL to
 *                         draw them from Philox4x32-10 keyed (seed; env_base+n, episode)
 *   obs_out    [N,obs_dim] f32
 */
int ppo_env_reset(int kind, double *state, int *elapsed, float *ep_ret, int *ep_len,
                  uint32_t *episode, const double *reset_draw, uint64_t seed,
                  uint64_t env_base, float *obs_out, int N, void *stream);

/*
 * One transition of N envs, fused with the per-actor bookkeeping of one iteration of
 * the rollout loop.  Replaces SyncVecEnv.vec_step (src/ppo.py:283-302), the env physics
 * behind env.step (gymnasium CartPoleEnv/AcrobotEnv + TimeLimit(500) +
 * RecordEpisodeStatistics; call sites src/env_cartpole.py:7-8, src/env_acrobot.py:7-8),
 * SyncVecEnv.env_reset (src/ppo.py:277-281) and the actor loop src/ppo.py:342-388:
 *   - done[n]  = terminated, trunc[n] = TimeLimit truncation (both may be 1)
 *   - if neither is set and is_last != 0: trunc[n] += 1 and the env is NOT reset
 *     (forced truncation at the last rollout step, src/ppo.py:352-356)
 *   - if done+trunc > 0: the terminal observation goes to term_obs[n,:] (:377-379) and,
 *     unless the truncation was forced, the env is reset and obs_out[n,:] holds the
 *     reset observation (:381-382)
 *   - otherwise term_obs[n,:] is left untouched
 *   action     [N] i32
 *   reward     [N] f32, done [N] i32, trunc [N] i32, obs_out [N,obs_dim] f32
 *   term_obs   [N,obs_dim] f32 or NULL
 *   ret_ring / len_ring [N,ring] f32/i32 + ring_cnt [N] i32: last-`ring` finished
 *              episode returns / lengths per env (RecordEpisodeStatistics deques read at
 *              src/ppo.py:404-417), or all NULL
 * Other arguments as in ppo_env_reset.
 */
int ppo_env_step(int kind, double *state, int *elapsed, float *ep_ret, int *ep_len,
                 uint32_t *episode, const int *action, const double *reset_draw,
                 uint64_t seed, uint64_t env_base, float *obs_out, float *term_obs,
                 float *reward, int *done, int *trunc, float *ret_ring, int *len_ring,
                 int *ring_cnt, int ring, int N, int is_last, void *stream);

/*
 * GAE(lambda) reverse-time scan, dense bootstrap values.
 * Replaces PPO.compute_advantages (src/ppo.py:452-488) when the bootstrap value of the
 * terminal observation is stored at the (t, n) where the flag was raised:
 *   end_value_dense [T,N] f32 (read only where trunc != 0)
 *   adv [T,N] f32 out;  ret [T,N] f32 out or NULL (adv + values, src/ppo.py:498)
 * Arithmetic per element is the reference's, one rounding per operation, so the result
 * is bit-identical to the reference for flags in {0,1}.
 */
int ppo_gae_dense(const float *rewards, const float *values, const float *end_value_dense,
                  const int *done, const int *trunc, double gamma, double lam, float *adv,
                  float *ret, int T, int N, void *stream);

/*
 * ppo_gae_dense reading the flags as two BITS per element instead of the two i32 words:
 *   flag_words [ceil(T/16), N] u32, ppo_gae_flag_words(T, N) words: bits 2k and 2k+1 of word q of column n are
 *   done and trunc of row 16 q + k (unused high bits of the last word: 0).  ppo_rollout_fused_packed leaves
 *   them beside its i32 done / trunc buffers, which stay the API's and the reference's format
 *   (src/ppo.py:393-394).
 * The same scan (src/ppo.py:452-488) with the same operations and roundings: bit-identical advantages.  12.25 B
 * of HBM traffic per element (r, v, flag bits read; adv written) instead of 20; end_value_dense is fetched per
 * row and only by the lanes whose flag word has that row's trunc bit.  Wide batches run the time-split kernel
 * (four waves per 64 columns, a quarter of the rows each, the one running value handed from wave to wave).
 * This is the form the trainer's device path uses between the whole-rollout kernel and the update.
 * Requires N * 4 < 2^32.
 */
size_t ppo_gae_flag_words(int T, int N);
int ppo_gae_dense_packed(const float *rewards, const float *values, const float *end_value_dense,
                         const uint32_t *flag_words, double gamma, double lam, float *adv, float *ret,
                         int T, int N, void *stream);

/* Bytes of device workspace ppo_gae_ragged needs for N columns. */
size_t ppo_gae_ragged_workspace_bytes(int N);

/*
 * GAE(lambda) with the reference's ragged, actor-major end_values[E] list: the exact
 * drop-in for PPO.compute_advantages(rewards, values, end_values, done_flags,
 * trunc_flags) (src/ppo.py:452-488), including ep_counts/cumsum indexing (:468-469), the
 * per-step index decrement by done+trunc (:480), negative-index wrap-around, and the
 * (1 - terminations) arithmetic when both flags are set (:476-479).
 * Where torch would raise IndexError (index outside [-E, E)) the kernel uses 0.0 and
 * sets *err_flag (i32, may be NULL) to 1 instead of faulting.
 */
int ppo_gae_ragged(const float *rewards, const float *values, const float *end_values,
                   long E, const int *done, const int *trunc, double gamma, double lam,
                   float *adv, int T, int N, void *workspace, size_t workspace_bytes,
                   int *err_flag, void *stream);

/*
 * ppo_gae_ragged without a workspace argument and without any allocation.  NO COUNTERPART IN THE
 * REFERENCE: it exists for callers that cannot call hipMalloc (code under HIP graph capture, tight
 * serving loops, harnesses that forbid allocation).  Same kernels, same results bit for bit; the
 * scratch (start index per column + one total per 256 columns) lives in a device array inside the
 * library, (PPO_GAE_RAGGED_NOALLOC_MAX_N + PPO_GAE_RAGGED_NOALLOC_MAX_N / 256) * 8 bytes = 2.1 MB
 * per GPU the library runs on, held until the process ends.
 *   - N > PPO_GAE_RAGGED_NOALLOC_MAX_N is rejected with hipErrorInvalidValue (nothing launched):
 *     use ppo_gae_ragged with a caller workspace for wider batches.
 *   - Call it with the device of `stream` current: the first call on a device looks up that
 *     device's copy of the array (hipGetSymbolAddress; make that first call outside graph capture).
 *   - Calls on one GPU share that GPU's array: calls that may overlap (different streams, different
 *     host threads) RACE.  Serialise them, or use ppo_gae_ragged with separate workspaces.
 *   - Nothing has to be cleared between calls: every scratch element a call reads is written by
 *     the same call first.
 * All other arguments, the error flag and the return value as in ppo_gae_ragged.
 */
#define PPO_GAE_RAGGED_NOALLOC_MAX_N 262144
int ppo_gae_ragged_noalloc(const float *rewards, const float *values, const float *end_values,
                           long E, const int *done, const int *trunc, double gamma, double lam,
                           float *adv, int T, int N, int *err_flag, void *stream);

/* Bytes of device workspace ppo_adv_stats / ppo_loss_fwd_bwd / ppo_loss_logits_fwd_bwd need for B samples. */
size_t ppo_loss_workspace_bytes(long B);

/*
 * {count, mean, M2 = sum (a - mean)^2} of adv[0..B) in fp64 -> stats3 (device, 3 f64).
 * Feeds the minibatch advantage normalisation of src/ppo.py:500-501
 * (mean, unbiased std).  Separate from the loss so that data-parallel ranks can merge
 * their triples before normalising.
 */
int ppo_adv_stats(const float *adv, long B, double *stats3, void *workspace,
                  size_t workspace_bytes, void *stream);

/*
 * The same statistics for the minibatch adv[idx[0 .. B-1]] without gathering it first (the index
 * slice of src/ppo.py:581-588 applied to `advantages`; :497-500): bit-identical to ppo_adv_stats on
 * the gathered copy.  idx [B] i64, every entry a valid row of adv (not checked).
 */
int ppo_adv_stats_gather(const float *adv, const long *idx, long B, double *stats3,
                         void *workspace, size_t workspace_bytes, void *stream);

/*
 * Fused PPO minibatch loss, forward and backward.  Replaces the ratio at
 * src/ppo.py:595 and PPO.compute_losses (src/ppo.py:490-527) plus the autograd backward
 * of both losses.
 *   x       [B] f32  new log-probs (ratio_input = 0) or probability ratios (= 1, the
 *                    signature compute_losses itself has)
 *   old_lp  [B] f32  old log-probs (ignored / may be NULL when ratio_input = 1)
 *   new_v, old_v, adv [B] f32
 *   stats3  device {count, mean, M2} used when norm_adv != 0 (may be NULL otherwise)
 *   out4    device f32[4] = policy_loss, value_loss, clip_frac, kl_div
 *   g_x     [B] f32  d policy_loss / d x  (NULL to skip);  g_v [B] f32 d value_loss / d new_v
 * Gradient tie rules are torch autograd's (minimum/maximum split 1/2-1/2, clamp passes
 * gradient on the closed interval).
 */
int ppo_loss_fwd_bwd(const float *x, const float *old_lp, const float *new_v,
                     const float *old_v, const float *adv, const double *stats3,
                     double clip, int norm_adv, int clip_va, int ratio_input, long B,
                     float *out4, float *g_x, float *g_v, void *workspace,
                     size_t workspace_bytes, void *stream);

/*
 * The same loss for a Discrete (Categorical) policy head, taking the head's raw logits: also folds
 * in what Agent.get_actions_and_values does with torch.distributions.Categorical on the update
 * path (src/ppo.py:186-189 logits -> distribution, :212 log_prob(actions), :214 entropy().mean())
 * and the ratio at :595, so log-softmax / probs / entropy tensors are never materialised.
 *   logits  [B, num_actions] f32 row-major, 2 <= num_actions <= 8;  actions [B] i32 in
 *           [0, num_actions): an action outside that range (torch raises) turns the four losses,
 *           and that sample's gradients, into NaN
 *   out5    device f32[5] = policy_loss, value_loss, clip_frac, kl_div, mean entropy
 *   g_logits_pl  [B, num_actions]  d policy_loss / d logits   (NULL to skip)
 *   g_logits_ent [B, num_actions]  d mean-entropy / d logits  (NULL to skip)
 *   g_v          [B]               d value_loss / d new_v     (NULL to skip)
 * Everything else as ppo_loss_fwd_bwd with ratio_input = 0.
 */
int ppo_loss_logits_fwd_bwd(const float *logits, int num_actions, const int *actions,
                            const float *old_lp, const float *new_v, const float *old_v,
                            const float *adv, const double *stats3, double clip, int norm_adv,
                            int clip_va, long B, float *out5, float *g_logits_pl,
                            float *g_logits_ent, float *g_v, void *workspace,
                            size_t workspace_bytes, void *stream);

/*
 * One-launch minibatch gather for the update phase: out[i] = in[idx[i]] for the observations,
 * actions, old log-probs, old values and advantages of the flattened rollout (M = T*N rows).
 * Replaces the five fancy-index gathers at src/ppo.py:582-588 (the shuffled index list of
 * :575-576 stays a device permutation made by the caller).
 *   idx [B] i64 (negative values wrap like torch/numpy); rows of states / actions are
 *   state_words / action_words 32-bit words wide (f32 observations; i32 Discrete or f32 Box
 *   actions).  Any of the five sources may be NULL to skip it.
 *   err_flag (i32, may be NULL) is set to 1 and the row zero-filled where torch would raise
 *   IndexError (index outside [-M, M)).
 */
int ppo_minibatch_gather(const int64_t *idx, long B, long M, const void *states, int state_words,
                         const void *actions, int action_words, const float *log_probs,
                         const float *values, const float *adv, void *mb_states, void *mb_actions,
                         float *mb_log_probs, float *mb_values, float *mb_adv, int *err_flag,
                         void *stream);

/*
 * The epoch's shuffled index list on the device: idx_out [n] i64 = a pseudo-random permutation of
 * 0 .. n-1.  Replaces `np.random.shuffle(indices)` (src/ppo.py:575-576) -- and torch.randperm, whose
 * 64-bit key sort costs more than a minibatch step at BASELINE configs[1].  Index i goes through a
 * keyed bijection of [0, 2^k), 2^k >= n (six xorshift / odd-multiply / add rounds modulo 2^k), applied
 * again while the result is >= n (cycle walking).  The round keys are Philox4x32-10 words of
 * (seed; stream_id): a different permutation for every stream_id (the caller counts epochs), the same
 * one on every device and for every sharding.  Not numpy's generator: no reference stream to match;
 * tests/test_gpu_shuffle.py checks that it is a permutation, equal to the oracle's restatement bit for
 * bit, and its statistics (position uniformity, neighbour independence, minibatch composition).
 * n <= 2^32.
 */
int ppo_shuffle_indices(long n, uint64_t seed, uint64_t stream_id, int64_t *idx_out, void *stream);

/*
 * Advantage statistics of EVERY minibatch of an epoch in one pass: stats3 [num_mb][3] f64 =
 * {count, mean, M2} (as ppo_adv_stats) of adv[idx[m * batch : (m + 1) * batch]], m = 0 .. num_mb - 1,
 * num_mb = ceil(n / batch) <= 8, where idx is the permutation ppo_shuffle_indices(n, seed, stream_id)
 * writes -- the minibatch slices of src/ppo.py:578-588, whose advantages compute_losses normalises with
 * their own mean / std (:501).  idx itself is not read: the permutation's bijection is run backwards,
 * element j belongs to minibatch inverse(j) / batch, so adv is streamed once, coalesced, instead of being
 * gathered once per minibatch (a 64-byte sector per 4-byte value).  f64 shifted sums in a fixed order
 * (deterministic); equal to ppo_adv_stats_gather on the same slices up to f64 summation order.
 * workspace: ppo_adv_stats_epoch_workspace_bytes() bytes, 8-byte aligned.
 */
size_t ppo_adv_stats_epoch_workspace_bytes(void);
int ppo_adv_stats_epoch(const float *adv, long n, uint64_t seed, uint64_t stream_id, long batch,
                        double *stats3, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Weight and bias gradient of y = x W^T + b for a small layer (M, K <= 64) over a huge batch:
 * dW[M,K] = sum_b g[b,m] x[b,k], db[M] = sum_b g[b,m] (db may be NULL).  Replaces the weight-gradient
 * GEMMs inside loss.backward() of the update phase (src/ppo.py:606-607, :614-615, :623-624) for the
 * dense network's Linear layers (:126-145); the batch is streamed once, partial sums are added in a
 * fixed order (deterministic).  g [B,M], x [B,K] f32 row-major; workspace of
 * ppo_linear_wgrad_workspace_bytes(B) bytes.
 */
size_t ppo_linear_wgrad_workspace_bytes(long B);
int ppo_linear_wgrad(const float *g, const float *x, long B, int M, int K, float *dW, float *db,
                     void *workspace, size_t workspace_bytes, void *stream);

/*
 * Categorical sampling for the rollout's policy step: for each row of logits[N,A]
 * draw an action with Philox4x32-10 keyed (seed; env_base+n, step / 4), word step % 4 (one block
 * serves four consecutive steps of an env), and return its log-probability.  Replaces torch.distributions.Categorical(logits).sample() and
 * .log_prob() at src/ppo.py:186-189,212 inside SyncVecEnv.rollout (:337).
 *   action [N] i32, logp [N] f32.  A <= 32.
 */
int ppo_categorical_sample(const float *logits, int A, uint64_t seed, uint64_t env_base,
                           uint64_t step, int *action, float *logp, int N, void *stream);

/* Number of floats in the packed dense-network parameter buffer ppo_rollout_fused reads
 * (layout in ppo_amd/csrc/rollout.hip; built by ppo_rollout_pack_params). */
size_t ppo_rollout_param_floats(int kind, int joint, int net_mode);

/*
 * Pack the dense policy / value network of the reference's Agent (src/ppo.py:124-145: pi_backbone
 * Linear(obs,64)-tanh-Linear(64,64)-tanh, optional va_backbone of the same shape, policy head
 * Linear(64,A), critic head Linear(64,1)) into the buffer ppo_rollout_fused reads: one launch,
 * straight from the nn.Linear tensors (weights row-major [out, in], f32), so that the rollout that
 * follows an optimiser step never waits for host work.
 *   kind       fixes obs (4 / 6) and A (2 / 3);  joint != 0: va_* are ignored (may be NULL)
 *   net_mode   PPO_NET_F32, PPO_NET_F16_SPLIT or PPO_NET_BF16X3 (the layouts differ)
 *   params_out ppo_rollout_param_floats(kind, joint, net_mode) floats
 *   split_diag device f32[3] or NULL: what decides whether PPO_NET_F16_SPLIT is as accurate as f32
 *              for this parameter set (w' = 2 log2(e) w is the value the split kernels split):
 *              [0] min over the backbone weight tensors of 8 sum|w'| / #{|w'| < 2^-3}: the ratio of the
 *                  mode's inherent 2^-22 relative error to the 2^-25 absolute floor of the low pieces
 *                  of small weights (f16 subnormals), for inputs of equal size;
 *              [1], [2] root-mean-square row norm of w' of the first / second backbone layer (max
 *                  over the backbones): the size of the pre-activations, where a 22-bit product is
 *                  less accurate than an f32 one.
 *              ppo_amd.ops.split_ok() holds the measured thresholds ([2] <= 8.5 is the one that
 *              matters: 2x the reference's orthogonal initialisation); use PPO_NET_BF16X3 (or PPO_NET_F32)
 *              outside them.
 */
int ppo_rollout_pack_params(int kind, int joint, int net_mode, const float *pi_w1, const float *pi_b1,
                            const float *pi_w2, const float *pi_b2, const float *va_w1,
                            const float *va_b1, const float *va_w2, const float *va_b2,
                            const float *policy_w, const float *policy_b, const float *critic_w,
                            const float *critic_b, float *params_out, float *split_diag,
                            void *stream);

/*
 * A whole T-step rollout of N envs in ONE launch: per time step the dense 64-64 tanh policy /
 * value network forward, Categorical sampling + log-prob, and the fused env transition of
 * ppo_env_step, with every [T,N] rollout buffer written in place.  Replaces the loop body of
 * SyncVecEnv.rollout (src/ppo.py:320-398) together with Agent.get_actions_and_values for the
 * dense network with a Discrete head (src/ppo.py:124-145,166-216) on CartPole / Acrobot.
 *   params      packed f32 network (ppo_rollout_param_floats), joint != 0: shared backbone
 *   sample_step0  Philox position of time step 0 (same stream as ppo_categorical_sample)
 *   obs_io      [N,obs_dim] f32: current observations in, carried observations out (:398)
 *   states [T,N,obs_dim] f32, actions [T,N] i32, log_probs/values/rewards [T,N] f32,
 *   done/trunc [T,N] i32, term_obs [T,N,obs_dim] f32 (written where done+trunc > 0)
 *   ev_dense [T,N] f32 or NULL: the critic's value of the terminal observation, written where
 *   trunc != 0 -- the end_value_dense input of ppo_gae_dense, i.e. what the reference computes
 *   with agent.get_values(end_states) (src/ppo.py:555) and reads where trunc != 0 (:478)
 *   reward_sum  device f64 or NULL: the sum of all T*N rewards is ADDED to it (the caller zeroes
 *   it) -- SyncVecEnv.total_return (src/ppo.py:318, :348); per-env f32 sums over time, f64 across envs
 * Env state arguments and rings as in ppo_env_step; row T-1 gets the forced truncation.
 * net_mode: one of the three PPO_NET_* modes; `params` must have been packed for that mode.
 * Values / log-probs agree with the PyTorch f32 forward of the same parameters to ~1e-6 (different
 * summation order than hipBLASLt; the split mode keeps 22 bits per operand), not bit for bit.
 */
int ppo_rollout_fused(int kind, double *state, int *elapsed, float *ep_ret, int *ep_len,
                      uint32_t *episode, const double *reset_draw, uint64_t seed,
                      uint64_t env_base, const float *params, int joint, int net_mode,
                      uint64_t sample_step0,
                      float *obs_io, float *states, int *actions, float *log_probs, float *values,
                      float *rewards, int *done, int *trunc, float *term_obs, float *ev_dense,
                      double *reward_sum, float *ret_ring, int *len_ring, int *ring_cnt, int ring,
                      int N, int T, void *stream);

/* ppo_rollout_fused that also writes flag_words (ppo_gae_flag_words(T, N) u32, layout as in
 * ppo_gae_dense_packed): one more store per env and sixteen steps on top of the 40 B per env-step of the
 * buffers above. */
int ppo_rollout_fused_packed(int kind, double *state, int *elapsed, float *ep_ret, int *ep_len,
                             uint32_t *episode, const double *reset_draw, uint64_t seed,
                             uint64_t env_base, const float *params, int joint, int net_mode,
                             uint64_t sample_step0,
                             float *obs_io, float *states, int *actions, float *log_probs, float *values,
                             float *rewards, int *done, int *trunc, float *term_obs, float *ev_dense,
                             uint32_t *flag_words, double *reward_sum, float *ret_ring, int *len_ring,
                             int *ring_cnt, int ring, int N, int T, void *stream);

/*
 * Episode statistics from the device rings.  Replaces the tail of SyncVecEnv.rollout
 * (src/ppo.py:404-417): np.concatenate of every env's return_queue / length_queue, np.max,
 * np.percentile(returns, [5, 50, 95]) and np.percentile(lengths, 50).
 * stats5 (device f64[5]) = {running max, p5, p50, p95 of the returns, median length}, "linear"
 * interpolation like np.percentile, exact (a radix select over the valid ring slots; nothing is
 * sorted).  Left untouched while no episode has finished, so initialise stats5 to NaN
 * (src/ppo.py:250-254); stats5[0] is a running maximum across calls (:408-410).
 * workspace: ppo_episode_stats_workspace_bytes() bytes of device memory, 8-byte aligned, ZERO-FILLED
 * by the caller once after allocating it; every call leaves it zero-filled for the next one (the
 * kernels clear what they have consumed, which saves a memset launch per call).  Do not share one
 * workspace between calls that may overlap.
 */
size_t ppo_episode_stats_workspace_bytes(void);
int ppo_episode_stats(const float *ret_ring, const int *len_ring, const int *ring_cnt, int ring,
                      int N, double *stats5, void *workspace, size_t workspace_bytes,
                      void *stream);

/*
 * The same statistics (src/ppo.py:404-417), same results bit for bit, for rings that only change by
 * episodes being appended between calls -- ppo_env_step / ppo_rollout_fused writing slot
 * ring_cnt % ring and incrementing ring_cnt, i.e. the reference's deque.append.  Only the slots
 * written since the previous call are visited (a 128-step rollout finishes a handful of episodes
 * per env, the rings hold 100); the two full passes of ppo_episode_stats run only on the first
 * call, when a percentile has moved to a different 12-bit key prefix, or for non-integer values.
 * state: ppo_episode_stats_state_bytes(N, ring) bytes of device memory, 16-byte aligned, ZERO-FILLED
 * by the caller once; it belongs to ONE set of rings (it holds each env's count and a copy of the
 * rings as of the last call, 8 * N * ring bytes).  Zero-fill it again whenever the rings were written
 * by anything else (a restored checkpoint, a count that went down); the next call then starts over
 * from the full rings.  Calls on one state must not overlap.
 */
size_t ppo_episode_stats_state_bytes(int N, int ring);
int ppo_episode_stats_incremental(const float *ret_ring, const int *len_ring, const int *ring_cnt,
                                  int ring, int N, double *stats5, void *state, size_t state_bytes,
                                  void *stream);

/*
 * The same statistics (src/ppo.py:404-417) for rings that hold only SMALL INTEGERS: returns that are integers in
 * [-512, 512] and lengths in [0, 512] -- the returns and lengths of CartPole-v1 / Acrobot-v1 (sums of +-1 rewards and
 * step counts under TimeLimit(500), src/env_cartpole.py:7-8, src/env_acrobot.py:7-8), i.e. everything the env kernels
 * of this library write.  Such a window is counted directly, one bin per integer, in a persistent histogram that is
 * updated from the slots written since the previous call (like ppo_episode_stats_incremental) and IS the sorted
 * window: the nine order statistics are nine positions in its prefix sum.  ONE launch per call -- no radix levels, no
 * pass that may or may not have work.  Same results as ppo_episode_stats bit for bit.
 * The promise is CHECKED, not trusted: a value that is not such an integer is counted for as long as it stays in the
 * window, and while there is one stats5 is set to NaN (all five; the running maximum restarts with the next good
 * call) instead of a wrong value.  What SyncVecEnv.rollout calls for the native envs.
 * state: as for ppo_episode_stats_incremental (same size, ppo_episode_stats_state_bytes; zero-filled once, and again
 * whenever the rings were rewritten by anything else), but the two forms keep different histograms in it: a state
 * block belongs to ONE of the two entry points -- zero-fill it when switching.
 * Data-parallel ranks add up the direct histograms (one all-reduce) and read the job's order statistics off the sum:
 * at byte offset ppo_episode_stats_ints_hist_offset() of the state, u32 [PPO_STATS_INTS_REPLICAS][PPO_STATS_INTS_RET_BINS +
 * PPO_STATS_INTS_LEN_BINS] -- partial histograms, kept apart so that the workgroups' updates do not queue on the same
 * words; their sum modulo 2^32 (a partial count may have wrapped below zero) is the window's histogram: word
 * PPO_STATS_INTS_RET_ZERO + r counts the returns equal to r, word PPO_STATS_INTS_RET_BINS + l the lengths equal to l.
 */
#define PPO_STATS_INTS_REPLICAS 8
#define PPO_STATS_INTS_RET_ZERO 512
#define PPO_STATS_INTS_RET_BINS 1280
#define PPO_STATS_INTS_LEN_BINS 768
#define PPO_STATS_INTS_LEN_MAX 512
size_t ppo_episode_stats_ints_hist_offset(void);
int ppo_episode_stats_incremental_ints(const float *ret_ring, const int *len_ring, const int *ring_cnt,
                                       int ring, int N, double *stats5, void *state, size_t state_bytes,
                                       void *stream);

/*
 * Job-wide episode statistics when the envs are sharded over data-parallel ranks.  The reference takes its
 * percentiles over EVERY env's deques (src/ppo.py:404-417); with one env shard per GPU the same numbers come from
 * a radix select whose histograms are added up over the ranks, one all-reduce per level, and narrowed by the
 * caller (ppo_amd/ppo.py: SyncVecEnv.job_wide_episode_stats).  Keys: k(x) = bits(x) ^ 0xffffffff for negative
 * floats, bits(x) | 0x80000000 otherwise (unsigned order = float order); lengths are counted as (float)length.
 *   - level 0: ppo_episode_stats_incremental keeps it.  At byte offset ppo_episode_stats_hist0_offset() of its
 *     state block: u32 [2][4096], counts of k >> 20 over the valid ring entries, returns then lengths.
 *   - levels 1 and 2: ppo_episode_hist_level ADDS to hist (device u32 [9 * 1024 + 1], zeroed by the caller) the
 *     counts of (k >> 10) & 1023 (level 1) or k & 1023 (level 2) over the valid ring entries whose higher key
 *     bits, k >> 20 (level 1) or k >> 10 (level 2), equal a listed prefix: row q < n_ret (<= 7) for returns
 *     under prefixes[q], row 7 + q, q < n_len (<= 2), for lengths under prefixes[7 + q]; prefixes: device u32[9].
 *     Level 1 also adds to hist[9 * 1024] a non-zero count if a matched value has any of its low 10 float bits set
 *     (zero on every rank: the 22 bits known after level 1 are the whole key -- integer-valued statistics).
 */
size_t ppo_episode_stats_hist0_offset(void);
int ppo_episode_hist_level(const float *ret_ring, const int *len_ring, const int *ring_cnt, int ring, int N,
                           int level, const uint32_t *prefixes, int n_ret, int n_len, uint32_t *hist,
                           void *stream);

/*
 * One minibatch of the update for the dense Agent with a Discrete head, fused: forward, loss,
 * backward and weight gradients in one pass over the samples.  Replaces, for one backbone and the
 * head(s) on it, Agent.get_actions_and_values on the minibatch (src/ppo.py:166-216, network
 * :124-145), the ratio (:595), compute_losses (:490-527), the loss assembly (:602-604 joint;
 * :611-612, :619-620 split networks) and loss.backward() (:606, :614, :622): `grads` receives
 *   d [ c_pi * policy_loss + c_v * value_loss - c_ent * entropy ] / d parameters
 * in the order w1 [64, obs], b1 [64], w2 [64, 64], b2 [64], then policy_w [A, 64], policy_b [A] if
 * heads & 1, then critic_w [64], critic_b [1] if heads & 2 (ppo_dense_update_grad_floats floats;
 * nn.Linear layouts), and out5 = {policy_loss, value_loss, clip_frac, kl_div, entropy} as
 * ppo_loss_logits_fwd_bwd defines them (entries of a head that is not given are left untouched).
 *   heads = 3: joint network.  Split networks: one call with the policy backbone and heads = 1
 *   (policy_loss, entropy, clip_frac, kl_div), one with the value backbone and heads = 2.
 *   w1 .. critic_b: the nn.Linear tensors (row-major [out, in]); policy_* / critic_* may be NULL
 *   for a head that is not given.  states [rows, obs] f32, actions [rows] i32, old_lp / old_v / adv
 *   [rows] f32: the flattened rollout buffers; idx [B] i64: the rows that form the minibatch (the
 *   slice of the shuffled indices, :575-588 -- the kernel gathers as it goes), or NULL for rows
 *   0 .. B-1.  stats3: statistics of the MINIBATCH's advantages as for ppo_loss_fwd_bwd.
 *   obs_dim 4 or 6, num_actions 2 or 3.
 * Arithmetic: net_mode = PPO_NET_F32: exact f32 products (f32 MFMA) everywhere.  PPO_NET_F16_SPLIT: the
 * two 64 x 64 products per sample (h2 = W2 h1 and dh1 = W2^T dz2) run as f16 MFMAs on two-piece splits
 * of both operands, as in ppo_rollout_fused (same validity limits: ppo_rollout_pack_params'
 * split_diag); dz2 is scaled per 32-sample tile by a power of two so that it fits f16's range
 * whatever 1 / B and the value error are.  The weight-gradient sums are exact-f32 products in both
 * modes.  tanh = 1 - 2 / (exp(2x) + 1) on the exp2 / rcp units (absolute error < 2e-7), per-sample
 * loss terms as in ppo_loss_logits_fwd_bwd, gradient sums per wave in f32 and across waves in f64
 * in a fixed order (run-to-run deterministic).  Agrees with PyTorch autograd of the same f32 network
 * to f32 summation-order accuracy.
 * workspace: ppo_dense_update_workspace_bytes() bytes, 8-byte aligned; not shared between calls
 * that may overlap.
 */
size_t ppo_dense_update_workspace_bytes(void);
size_t ppo_dense_update_grad_floats(int obs_dim, int num_actions, int heads);
int ppo_dense_update_grads(int obs_dim, int num_actions, int heads, int net_mode, const float *w1,
                           const float *b1, const float *w2, const float *b2, const float *policy_w,
                           const float *policy_b, const float *critic_w, const float *critic_b,
                           const long *idx, const float *states, const int *actions,
                           const float *old_lp, const float *old_v, const float *adv,
                           const double *stats3, double clip,
                           int norm_adv, int clip_va, long B, double c_pi, double c_v, double c_ent,
                           float *out5, float *grads, void *workspace, size_t workspace_bytes,
                           void *stream);

/* ---- gradient clipping + Adam step of one optimiser group (csrc/optim.hip) --------------------
 * Replaces, per minibatch, nn.utils.clip_grad_norm_(agent.parameters(), max_grad_norm) followed by
 * optimizer.step() (src/ppo.py:608-609 joint; :615-616 and :625-626 split networks; the optimisers
 * are optim.Adam(lr, eps=1e-5), :732-733) for the dense agents: ONE launch instead of PyTorch's 17.
 *   params / exp_avg / exp_avg_sq: HOST arrays of n_tensors (<= 16) device pointers -- the group's
 *   parameter tensors and their Adam moments, numel[k] f32 each; grads: the group's gradients in the
 *   same order as ONE flat device buffer of sum(numel) f32 (what ppo_dense_update_grads writes),
 *   scaled in place by the clip coefficient like clip_grad_norm_ does.
 *   total_norm = sqrt(sum g^2) (f64 sum); coef = min(max_grad_norm / (total_norm + 1e-6), 1)
 *   (max_grad_norm < 0: no clipping); then torch's Adam (no weight decay, no amsgrad):
 *   m += (1 - beta1)(g - m);  v = beta2 v + (1 - beta2) g^2;
 *   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps),  step = 1, 2, ...
 *   in f32, one rounding per operation.  One workgroup: sum(numel) <= 2^22.
 */
int ppo_clip_adam_step(int n_tensors, float *const *params, float *const *exp_avg,
                       float *const *exp_avg_sq, const long *numel, float *grads,
                       double max_grad_norm, double lr, double beta1, double beta2, double eps,
                       long step, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PPO_AMD_H */
