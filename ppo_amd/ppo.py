"""Python surface of the MI355X-native PPO path: ScaledBeta, Agent, SyncVecEnv, PPO.

Same class / method / attribute names and argument meaning as the reference's src/ppo.py
(cited per item), written from scratch around libppo_amd.so:

* env stepping, auto-reset, terminal-obs capture, episode statistics  -> ppo_env_step
* GAE(lambda)                                                          -> ppo_gae_dense / _ragged
* ratio + clipped surrogate + value loss + their gradients            -> ppo_loss_fwd_bwd
* the small policy/value MLP (and the Atari CNN) stay on PyTorch-ROCm (north star).

Out of scope on purpose: GIF rendering, wandb.  The flags are
accepted; metrics keep the reference's names and go to `PPO.last_metrics` (+ a JSONL file
when PPO_AMD_METRICS is set).
"""
from __future__ import annotations

import json
import os
import time
from typing import Callable

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from . import dist as ppo_dist
from . import envs as ppo_envs
from . import ops, spaces

try:  # optional, never required
    import tqdm as _tqdm
except Exception:  # pragma: no cover
    _tqdm = None


# ----------------------------------------------------------------------------------------
class ScaledBeta(torch.distributions.Distribution):
    """Beta on [low, high] parameterised by mode and precision (reference src/ppo.py:18-56):
    alpha = mode * precision + 1, beta = precision + 2 - alpha; log-prob and entropy are
    summed over the action dimension."""

    def __init__(self, modes, precisions, lows, highs):
        super().__init__(validate_args=False)
        self.modes, self.precisions, self.lows, self.highs = modes, precisions, lows, highs
        self.alphas = modes * precisions + 1.0
        self.betas = precisions + 2.0 - self.alphas
        self.diffs = highs - lows
        self.log_diffs = torch.log(self.diffs)
        self.epsilon = 1e-6
        self.unit_beta_dist = torch.distributions.Beta(self.alphas, self.betas)

    def sample(self):
        return self.unit_beta_dist.sample() * self.diffs + self.lows

    def log_prob(self, variates):
        unit = (variates - self.lows) / (self.diffs + self.epsilon)
        return torch.sum(self.unit_beta_dist.log_prob(unit) - self.log_diffs, dim=1)

    def entropy(self):
        return torch.sum(self.unit_beta_dist.entropy() + self.log_diffs, dim=1)


# ----------------------------------------------------------------------------------------
class _Linear(nn.Linear):
    """nn.Linear (same parameters, same state_dict) whose weight / bias gradients on large CUDA
    batches come from ppo_linear_wgrad: the BLAS back ends run those batch-size-deep reductions at
    2-4 ms per layer for a 2M-sample minibatch, 70 % of an update."""

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


def _ortho(layer, std=float(np.sqrt(2))):
    nn.init.orthogonal_(layer.weight, std)  # reference src/ppo.py:87-89 (bias keeps torch default)
    return layer


class Agent(nn.Module):
    """Policy/value network with the reference's architecture and init (src/ppo.py:58-216):
    dense 64-64 tanh (or Nature-CNN -> 512) backbone(s), policy head gain 0.01, critic gain 1."""

    def __init__(self, state_space, action_space, conv_net, joint_net, device):
        super().__init__()
        self.action_space, self.state_space = action_space, state_space
        self.conv_net, self.joint_net = conv_net, joint_net
        self.discrete = spaces.is_discrete(action_space)
        if not self.discrete and not spaces.is_box(action_space):
            raise TypeError("action_space must be Discrete-like (.n) or Box-like (.low/.high/.shape)")
        self.image_states = len(state_space.shape) >= 3
        n_out = action_space.n if self.discrete else 2 * action_space.shape[0]
        width = 512 if conv_net else 64
        self.pi_backbone = self._backbone()
        if not joint_net:
            self.va_backbone = self._backbone()
        head = nn.Linear if conv_net else _Linear
        self.policy = _ortho(head(width, n_out), 0.01)
        self.critic = _ortho(head(width, 1), 1.0)
        self.device = device
        self.to(self.device, dtype=torch.float32)
        self._bounds = None

    def _backbone(self):
        shape = self.state_space.shape
        if self.conv_net:  # reference src/ppo.py:93-103
            return nn.Sequential(
                _ortho(nn.Conv2d(shape[-1], 32, 8, stride=4)), nn.ReLU(),
                _ortho(nn.Conv2d(32, 64, 4, stride=2)), nn.ReLU(),
                _ortho(nn.Conv2d(64, 64, 3, stride=1)), nn.ReLU(),
                nn.Flatten(),
                _ortho(nn.Linear(64 * 7 * 7, 512)), nn.ReLU())
        return nn.Sequential(  # reference src/ppo.py:126-131
            _ortho(_Linear(shape[0], 64)), nn.Tanh(),
            _ortho(_Linear(64, 64)), nn.Tanh())

    def permute_states(self, states):
        return torch.permute(states, (0, 3, 1, 2))  # NHWC -> NCHW, reference src/ppo.py:147-148

    def _prep(self, states):
        return self.permute_states(states) if self.image_states else states

    def get_values(self, states):
        """reference src/ppo.py:150-164"""
        states = self._prep(states)
        hidden = self.pi_backbone(states) if self.joint_net else self.va_backbone(states)
        return torch.flatten(self.critic(hidden))

    def policy_and_values(self, states):
        """Raw policy head output [B, n_out] and values [B] (first half of
        get_actions_and_values, reference src/ppo.py:172-182)."""
        states = self._prep(states)
        pi_hidden = self.pi_backbone(states)
        va_hidden = pi_hidden if self.joint_net else self.va_backbone(states)
        return self.policy(pi_hidden), self.critic(va_hidden).flatten()

    def _action_bounds(self, device):
        if self._bounds is None or self._bounds[0].device != device:
            n = self.action_space.shape[0]
            lows, highs = self.action_space.low, self.action_space.high
            if not isinstance(lows, np.ndarray):  # scalar bounds, reference src/ppo.py:196-198
                lows = lows * np.ones((n,), dtype=np.float32)
                highs = highs * np.ones((n,), dtype=np.float32)
            self._bounds = (torch.tensor(lows, device=device), torch.tensor(highs, device=device))
        return self._bounds

    def distribution(self, policy_output):
        if self.discrete:
            return torch.distributions.Categorical(logits=policy_output)
        n = self.action_space.shape[0]
        lows, highs = self._action_bounds(policy_output.device)
        modes = 0.5 * (torch.clip(policy_output[:, :n], min=-1.0, max=1.0) + 1)  # reference :204
        precisions = torch.exp(policy_output[:, n:])                              # reference :205
        return ScaledBeta(modes, precisions, lows, highs)

    def get_actions_and_values(self, states, actions):
        """reference src/ppo.py:166-216 -> (actions, log_probs, values, mean entropy)"""
        policy_output, values = self.policy_and_values(states)
        dist = self.distribution(policy_output)
        if actions is None:
            actions = dist.sample().to(dtype=torch.int32 if self.discrete else torch.float32)
        return actions, dist.log_prob(actions), values, dist.entropy().mean()


_BATCH_TENSORS = ("state", "elapsed", "ep_ret", "ep_len", "episode", "ret_ring", "len_ring", "ring_cnt")


# ----------------------------------------------------------------------------------------
class SyncVecEnv:
    """Vectorised env + rollout collector with the reference's contract (src/ppo.py:218-417).

    If `env_fn()` yields a ppo_amd.envs.DeviceEnv (CartPole / Acrobot) all `num_envs` envs
    live in HBM and one ppo_env_step launch per time step does stepping, flags, forced
    last-step truncation, terminal-obs capture, auto-reset and episode statistics.  Any
    other gym.Env-protocol object keeps its own host stepping (Breakout / Galaxian / Ant in
    the reference) and only GAE and the loss run on the GPU kernels.
    """

    def __init__(self, env_fn: Callable, num_envs: int, steps_per_env: int, render_every: int,
                 render_fps: float, agent: Agent, seed: int | None = None, env_base: int = 0):
        self.num_envs, self.steps_per_env = num_envs, steps_per_env
        self.global_steps = 0
        self.can_record = render_every > 0
        if self.can_record:
            print("ppo_amd: GIF rendering is out of scope for device-resident envs; render_every ignored")
        self.render_every, self.render_fps = (render_every if render_every > 0 else 1), render_fps
        self.agent = agent
        # episode statistics (reference src/ppo.py:250-254): host values, or a device tensor that is
        # fetched on first read (see the properties below) so that a rollout never waits on the host
        nan = np.float32(np.nan)
        self._stats_host = {"max": nan, "lower": nan, "median": nan, "upper": nan, "length": nan}
        self._stats_dev = None
        self._stats_pending = False
        self._stats5 = None
        self._stats_state = None
        self._rsum_pool, self._rsum_next = torch.empty(0), 0
        self._total_return = np.float32(0.0)
        self._local_max = np.float32(np.nan)       # host-stepped envs under data parallelism: this rank's own running max
        self._end_states = None
        self.state_space, self.action_space = agent.state_space, agent.action_space
        self.action_dtype = torch.int32 if spaces.is_discrete(self.action_space) else torch.float32
        self.device = agent.device
        self.fused_sampling = True  # device path: Categorical sample + log_prob in one kernel
        # device path, dense net + Discrete head: the whole T-step rollout (policy forward, sampling,
        # env transition) in ONE kernel launch.  False = PyTorch MLP + one env kernel per time step.
        self.fused_rollout = True
        # how the whole-rollout kernel multiplies inside the network (include/ppo_amd.h): exact f32
        # MFMA products (ops.NET_F32), f16 MFMAs on two-piece splits (ops.NET_F16_SPLIT, ~2x faster,
        # 22 bits per operand), bf16 MFMAs on three-piece splits (ops.NET_BF16X3: operands exact, f32-grade
        # at every weight scale), or -- the default -- the f16-split kernels whenever the pack kernel's
        # split_diag says they are as accurate as f32 for the current parameters, bf16x3 otherwise
        # (ops.NET_AUTO).  PPO_AMD_NET = auto | f32 | f16_split | bf16x3.
        self.net_mode = {"auto": ops.NET_AUTO, "f32": ops.NET_F32, "f16_split": ops.NET_F16_SPLIT,
                         "bf16x3": ops.NET_BF16X3}[os.environ.get("PPO_AMD_NET", "auto")]
        self.net_mode_used = None    # what the last whole-rollout launch ran (NET_F32 / NET_F16_SPLIT / NET_BF16X3)
        self._pack = None            # parameters packed for the rollout kernel, see _net()
        self.wave_envs = 0           # tests / tuning: force 32 or 64 envs per wave (0 = the library chooses by N)
        self._rollouts = 0
        self._bufset = None           # see _alloc
        self._params = None           # (module._parameters dict, name) per parameter of the agent, see prepack

        first = env_fn()
        self.native = isinstance(first, ppo_envs.DeviceEnv)
        if self.native:
            self.batch = ppo_envs.DeviceEnvBatch(first.kind, num_envs, seed=seed if seed is not None else first.seed,
                                                 env_base=env_base)
            self.kdev = self.batch.device           # where the env kernels run (always HIP)
            self.envs = [first]
            self._term_obs = None
            self._ev_dense = None
            self._flag_words = None                    # [ceil(T/16),N]: two flag bits per element, what the packed GAE kernel reads
        else:
            self.batch, self.kdev = None, None
            self.envs = [first] + [env_fn() for _ in range(num_envs - 1)]
        self.t_states = self.vec_reset()

    def close(self):
        self.envs = None
        self.batch = None

    # -- reset / step ---------------------------------------------------------------------
    def vec_reset(self) -> torch.Tensor:
        """reference src/ppo.py:269-275"""
        if self.native:
            return self.batch.reset().to(self.device)
        states = torch.zeros((self.num_envs, *self.state_space.shape), dtype=torch.float32, device=self.device)
        for i, env in enumerate(self.envs):
            states[i] = torch.as_tensor(np.asarray(env.reset()[0]), dtype=torch.float32)
        return states

    def env_reset(self, env_id: int):
        """reference src/ppo.py:277-281 (host envs only: device envs reset inside the step kernel)"""
        if self.native:
            raise RuntimeError("device-resident envs auto-reset inside ppo_env_step; env_reset is host-path only")
        return torch.as_tensor(np.asarray(self.envs[env_id].reset()[0]), dtype=torch.float32, device=self.device)

    def vec_step(self, actions: torch.Tensor):
        """reference src/ppo.py:283-302 -> (states, rewards, done_flags i32, trunc_flags i32).
        Device envs: the returned states are the NEXT observations, i.e. already the reset
        observation for envs that finished (the reference does that two lines later, :381-382)."""
        N = self.num_envs
        if self.native:
            self._flush_stats()
            d = self.kdev
            obs = torch.empty((N, self.batch.obs_dim), dtype=torch.float32, device=d)
            rew = torch.empty(N, dtype=torch.float32, device=d)
            done = torch.empty(N, dtype=torch.int32, device=d)
            trunc = torch.empty(N, dtype=torch.int32, device=d)
            act = actions.to(d, torch.int32).contiguous()
            self.batch.step(act, obs, None, rew, done, trunc, is_last=False)
            return tuple(x.to(self.device) for x in (obs, rew, done, trunc))
        acts = actions.cpu().numpy()
        states = np.zeros((N, *self.state_space.shape), dtype=np.float32)
        rewards = np.zeros(N, dtype=np.float32)
        dones = np.zeros(N, dtype=np.int32)
        truncs = np.zeros(N, dtype=np.int32)
        for i, env in enumerate(self.envs):
            states[i], rewards[i], dones[i], truncs[i], _ = env.step(np.squeeze(acts[i]))
        dev = self.device
        return (torch.from_numpy(states).to(dev), torch.from_numpy(rewards).to(dev),
                torch.from_numpy(dones).to(dev), torch.from_numpy(truncs).to(dev))

    # -- rollout --------------------------------------------------------------------------
    def _alloc(self, dev, reuse: bool = False):
        """The rollout buffers (reference src/ppo.py:306-314 allocates them anew every rollout).
        reuse (device rollouts): ONE set, overwritten by every rollout, instead of seven allocations per
        rollout -- ~20 us of host time in front of the rollout launch, which is inside the reference's
        env_steps_per_sec bracket, and the same addresses every time: the GAE kernel right behind the
        rollout finds more of its input in the Infinity Cache (29.0 us; 29.4 with fresh allocations, 31.9
        with two alternating sets).  A tensor taken from an earlier rollout is overwritten by the next one:
        clone what has to outlive it."""
        T, N = self.steps_per_env, self.num_envs
        if reuse:
            key = (T, N, str(dev))
            slot = self._bufset
            if slot is not None and slot[0] == key:
                (self.states, self.actions, self.rewards, self.done_flags, self.trunc_flags, self.values,
                 self.log_probs) = slot[1]
                return
        self.states = torch.empty((T, N, *self.state_space.shape), dtype=torch.float32, device=dev)
        self.actions = torch.empty((T, N, *self.action_space.shape), dtype=self.action_dtype, device=dev)
        self.rewards = torch.empty((T, N), dtype=torch.float32, device=dev)
        self.done_flags = torch.empty((T, N), dtype=torch.int32, device=dev)
        self.trunc_flags = torch.empty((T, N), dtype=torch.int32, device=dev)
        self.values = torch.empty((T, N), dtype=torch.float32, device=dev)
        self.log_probs = torch.empty((T, N), dtype=torch.float32, device=dev)
        if reuse:
            self._bufset = (key, (self.states, self.actions, self.rewards, self.done_flags,
                                                       self.trunc_flags, self.values, self.log_probs))

    def rollout(self):
        """reference src/ppo.py:304-417"""
        with torch.no_grad():
            if self.native:
                self._rollout_device()
            else:
                self._rollout_host()
        self._rollouts += 1
        self.states.requires_grad = True  # reference src/ppo.py:400

    def _rollout_device(self):
        T, N, b = self.steps_per_env, self.num_envs, self.batch
        self._flush_stats()                            # statistics of the previous rollout, before the rings change
        kdev, adev = self.kdev, torch.device(self.device)
        same = adev.type == "cuda"
        self._alloc(kdev, reuse=same)
        if self._term_obs is None or self._term_obs.shape[0] != T:
            self._term_obs = torch.empty((T, N, b.obs_dim), dtype=torch.float32, device=kdev)
        fused = self.fused_sampling and same and self.agent.discrete
        whole = (self.fused_rollout and same and self.agent.discrete and not self.agent.conv_net
                 and self.agent.action_space.n == ppo_envs.NUM_ACTIONS[b.kind])
        self._ev_ready = False
        if whole:
            if self._ev_dense is None or self._ev_dense.shape != self.rewards.shape:
                self._ev_dense = torch.empty_like(self.rewards)
            if self._flag_words is None or self._flag_words.shape != ((T + 15) // 16, N):
                self._flag_words = ops.gae_flag_words(T, N, kdev)
            carry = self.t_states.contiguous()         # obs_io: read at t = 0, overwritten with the carried obs
            # reward-sum cell of this rollout (read lazily later): cells come zeroed 64 at a time, so
            # that a rollout does not pay for a fill launch of its own
            if self._rsum_next >= self._rsum_pool.numel():
                self._rsum_pool, self._rsum_next = torch.zeros(64, dtype=torch.float64, device=kdev), 0
            rsum = self._rsum_pool[self._rsum_next:self._rsum_next + 1]
            self._rsum_next += 1
            params, mode = self._net()
            ops.rollout_fused(b, params, self.agent.joint_net, self._rollouts * T, carry,
                              self.states, self.actions, self.log_probs, self.values, self.rewards,
                              self.done_flags, self.trunc_flags, self._term_obs, self._ev_dense, T,
                              net_mode=mode, reward_sum=rsum, wave_envs=self.wave_envs, flag_words=self._flag_words)
            self.net_mode_used = mode
            self._ev_ready = True                      # bootstrap values already written by the kernel
        else:
            carry = torch.empty((N, b.obs_dim), dtype=torch.float32, device=kdev)
            self.states[0].copy_(self.t_states)
        for t in range(0 if whole else T):
            obs = self.states[t] if same else self.states[t].to(adev)
            if fused:
                logits, values = self.agent.policy_and_values(obs)
                ops.categorical_sample(logits, b.seed, b.env_base, self._rollouts * T + t,
                                       action_out=self.actions[t], logp_out=self.log_probs[t])
                self.values[t].copy_(values)
            else:
                a, lp, v, _ = self.agent.get_actions_and_values(obs, actions=None)
                self.actions[t].copy_(a)
                self.log_probs[t].copy_(lp)
                self.values[t].copy_(v)
            nxt = self.states[t + 1] if t + 1 < T else carry
            b.step(self.actions[t], nxt, self._term_obs[t], self.rewards[t], self.done_flags[t],
                   self.trunc_flags[t], is_last=(t == T - 1))
        self.global_steps += T * N
        self.t_states = carry if same else carry.to(adev)
        self._end_states = self._end_n = self._end_t = None   # built on first access (end_states property)
        # device scalar, fetched on read; the whole-rollout kernel sums the rewards itself
        self._total_return = rsum[0] if whole else self.rewards.sum(dtype=torch.float32)
        if not same:
            self._locate_end_states()
            for name in ("states", "actions", "rewards", "done_flags", "trunc_flags", "values", "log_probs",
                         "_end_states"):
                setattr(self, name, getattr(self, name).to(adev))
        # The statistics kernels are queued lazily -- before the next kernel that touches the rings, or
        # when a statistic is read: the caller's GAE then runs right behind the rollout kernel, while the
        # rows it starts with (the last ones written) are still in the 256 MiB Infinity Cache, instead
        # of behind 150 MB of ring reads.  Same values: only rollout / step kernels change the rings.
        self._stats_pending = True

    def prepack(self):
        """Queue the pack kernel(s) for the current parameters if they changed since the last pack
        (torch bumps a tensor's _version on every in-place update, i.e. on every optimiser step); no
        host synchronisation.  PPO.train_step calls this right behind the last optimizer.step() of an
        update, so that the next rollout starts with its parameters packed.  Returns the device f32[3]
        split_diag still to be read (ops.NET_AUTO), else None."""
        if self._params is None:                           # walking the module tree costs ~10 us per call: keep the slots
            self._params = [(m._parameters, n) for m in self.agent.modules() for n, v in m._parameters.items() if v is not None]
        # (_version: every in-place torch update; param_generation: every ppo_clip_adam_step; data_ptr: a parameter
        # re-created (the slot is looked up again each time), moved by module.to() or rewritten through `.data =`,
        # none of which bump _version)
        key = (self.net_mode, ops.param_generation()) + tuple([(p._version, p.data_ptr()) for p in
                                                               [d[n] for d, n in self._params]])
        pk = self._pack
        if pk is not None and pk["key"] == key:
            return pk["diag"] if pk["ok"] is None else None
        if pk is None or pk["mode"] != self.net_mode:
            pk = self._pack = {"mode": self.net_mode, "bufs": {}, "diag": None}
            if self.net_mode == ops.NET_AUTO:
                pk["diag"] = torch.empty(3, dtype=torch.float32, device=self.kdev)
        pk["key"] = key
        modes = (ops.NET_F16_SPLIT, ops.NET_AUTO_FALLBACK) if self.net_mode == ops.NET_AUTO else (self.net_mode,)
        for m in modes:                                    # AUTO: both layouts (two 5 us launches), decided below
            pk["bufs"][m] = ops.pack_dense_agent(self.agent, m, out=pk["bufs"].get(m),
                                                 diag_out=pk["diag"] if m == ops.NET_F16_SPLIT else None)
        pk["ok"] = None if self.net_mode == ops.NET_AUTO else True
        return pk["diag"]

    def note_split_diag(self, diag):
        """The host values of prepack()'s split_diag (read by the caller together with its other scalars)."""
        self._pack["ok"] = ops.split_ok(diag)

    def _net(self):
        """(packed parameters, network mode) for the next whole-rollout launch."""
        pending = self.prepack()
        if pending is not None:                            # not read by the caller: one small device->host copy
            self.note_split_diag(pending.tolist())
        pk = self._pack
        if self.net_mode != ops.NET_AUTO:
            return pk["bufs"][self.net_mode], self.net_mode
        mode = ops.NET_F16_SPLIT if pk["ok"] else ops.NET_AUTO_FALLBACK
        return pk["bufs"][mode], mode

    def _locate_end_states(self):
        """end_states: terminal observations, actor-major / time-ascending (reference :401-402).
        Costs one host sync (the count E), so it is built only when somebody asks for it."""
        ended = (self.done_flags.to(self.kdev) + self.trunc_flags.to(self.kdev)) > 0
        nt = ended.t().nonzero()
        self._end_n, self._end_t = nt[:, 0], nt[:, 1]
        self._end_states = self._term_obs[self._end_t, self._end_n]

    @property
    def end_states(self):
        if self._end_states is None and self.native:
            self._locate_end_states()
        return self._end_states

    @end_states.setter
    def end_states(self, value):
        self._end_states = value

    @property
    def total_return(self):
        if isinstance(self._total_return, torch.Tensor):
            self._total_return = np.float32(self._total_return.item())
        return self._total_return

    @total_return.setter
    def total_return(self, value):
        self._total_return = value

    def _flush_stats(self):
        if self._stats_pending:
            self._stats_pending = False
            self._device_episode_stats(incremental=True)

    def _device_episode_stats(self, incremental: bool = False):
        """reference src/ppo.py:407-417 on the device rings, without a host round trip: running max,
        5/50/95 percentiles (np.percentile 'linear') of the last <=100 returns of every env, median
        length -- a radix select, three launches on the rollout's stream.  `incremental` (what a rollout
        uses): ppo_episode_stats_incremental(_ints), which visits only the ring slots written since its last
        call and needs the rings to have changed by appended episodes only (the env kernels);
        otherwise ppo_episode_stats, a full scan of whatever the rings hold.  Same results.
        The result stays a device tensor until one of the statistics attributes is read."""
        b = self.batch
        lib = ops._lib.load()
        if self._stats5 is None:
            self._stats5 = torch.full((5,), float("nan"), dtype=torch.float64, device=b.device)
            self._stats_ws = torch.zeros(lib.ppo_episode_stats_workspace_bytes(), dtype=torch.uint8, device=b.device)
        rings = (b.ret_ring.data_ptr(), b.len_ring.data_ptr(), b.ring_cnt.data_ptr(), b.ring, b.num_envs)
        if incremental:
            if self._stats_state is None:
                self._stats_state = torch.zeros(lib.ppo_episode_stats_state_bytes(b.num_envs, b.ring), dtype=torch.uint8,
                                                device=b.device)
            # the rings are written by the env kernels only: returns are sums of +-1 rewards, lengths step counts
            # under TimeLimit(500) -- small integers, for which the select ends after level 1 (no third launch;
            # a value that is not one would turn the statistics into NaN, not into a wrong number)
            ops._lib.check(lib.ppo_episode_stats_incremental_ints(*rings, self._stats5.data_ptr(), self._stats_state.data_ptr(),
                                                                  self._stats_state.numel(), ops._stream()),
                           "ppo_episode_stats_incremental_ints")
        else:
            ops._lib.check(lib.ppo_episode_stats(*rings, self._stats5.data_ptr(), self._stats_ws.data_ptr(),
                                                 self._stats_ws.numel(), ops._stream()), "ppo_episode_stats")
        self._stats_dev = self._stats5

    def invalidate_episode_stats_state(self):
        """The rings were written by something other than the env kernels (a restored checkpoint, a test):
        the next incremental statistics call starts over from the full rings."""
        if self._stats_state is not None:
            self._stats_state.zero_()

    def _stat(self, key):
        self._flush_stats()
        if self._stats_dev is not None:
            if ppo_dist.is_dist():                      # this rank's shard only: the logged statistics are job-wide,
                self._stats_dev = None                  # made by job_wide_episode_stats (every rank, once per update)
            else:                                       # one device->host copy for all five values
                self._note_stats(self._stats_dev.cpu().numpy())
        return self._stats_host[key]

    def _note_stats(self, v):
        self._stats_dev = None
        self._stats_host = {"max": np.float32(v[0]), "lower": np.float64(v[1]), "median": np.float64(v[2]),
                            "upper": np.float64(v[3]), "length": np.float64(v[4])}

    def pending_device_scalars(self):
        """(f64 device tensor or None, callback): the episode statistics and the reward sum of the last rollout
        that still live on the device, for a caller that copies scalars of its own to the host anyway
        (PPO.train_step: ONE device->host copy per update instead of three); the callback takes the host values."""
        self._flush_stats()
        parts, takers = [], []
        if self._stats_dev is not None:
            parts.append(self._stats_dev)
            takers.append((5, self._note_stats))
        if isinstance(self._total_return, torch.Tensor) and self._total_return.dtype == torch.float64:
            parts.append(self._total_return.reshape(1))
            takers.append((1, lambda v: setattr(self, "_total_return", np.float32(v[0]))))
        if not parts:
            return None, lambda host: None

        def take(host):
            at = 0
            for n, fn in takers:
                fn(np.asarray(host[at:at + n], dtype=np.float64))
                at += n
        return (parts[0] if len(parts) == 1 else torch.cat(parts)), take

    @property
    def local_max_ep_return(self):
        """Running max over THIS rank's envs only (max_ep_return is job-wide under data parallelism)."""
        if not ppo_dist.is_dist() or not self.native:
            return self._local_max if (ppo_dist.is_dist() and not self.native) else self._stat("max")
        self._flush_stats()
        return np.float32(self._stats5[0].item()) if self._stats5 is not None else np.float32(np.nan)

    # -- job-wide statistics under data parallelism -------------------------------------------------
    def job_wide_episode_stats(self) -> float:
        """Data parallel: the reference's statistics tail (src/ppo.py:404-417) over the deques of ALL envs of the
        job, not of this rank's shard -- running max, 5/50/95 percentiles of the returns, median length -- and the
        job's reward sum of the last rollout (:318, :348), which it returns.  EVERY rank must call it, once per
        update (collectives: one gather of three floats per rank, then ONE sum of the ranks' direct histograms,
        8 KB, read identically on every rank's host).  Exact: the same nine order statistics ppo_episode_stats
        selects, np.percentile's "linear" interpolation."""
        if not self.native:
            return self._job_wide_host_stats()
        lib = ops._lib.load()
        b = self.batch
        self._flush_stats()
        dev = b.device
        tr = self._total_return
        tr = tr.detach().to(torch.float64).reshape(1) if isinstance(tr, torch.Tensor) else torch.tensor([float(tr)], dtype=torch.float64)
        own_max = self._stats5[0:1] if self._stats5 is not None else torch.full((1,), float("nan"), dtype=torch.float64, device=dev)
        # (IntsState of csrc/stats.hip: the count of window entries that are not small integers is the i64 at byte 8)
        own_bad = (self._stats_state[8:16].view(torch.int64).to(torch.float64) if self._stats_state is not None
                   else torch.zeros(1, dtype=torch.float64, device=dev))
        pair = torch.cat([torch.nan_to_num(own_max, nan=-float("inf")).to(dev), tr.to(dev), own_bad.to(dev)])
        rows = ppo_dist.gather_rows(pair, "ep_max_and_reward_sum").cpu().numpy()
        job_max, job_reward, job_bad = float(rows[:, 0].max()), float(rows[:, 1].sum()), float(np.abs(rows[:, 2]).sum())
        self._stats_dev = None
        if self._stats_state is None:                      # no rollout yet
            return job_reward
        # the incremental statistics keep the window as a DIRECT histogram (one bin per integer return / length,
        # ppo_episode_stats_incremental_ints): the job's window is the sum of the ranks' histograms, and the nine order
        # statistics are nine positions in its prefix sum
        RB, R0 = _lib_const("PPO_STATS_INTS_RET_BINS"), _lib_const("PPO_STATS_INTS_RET_ZERO")
        LB = _lib_const("PPO_STATS_INTS_LEN_BINS")
        off = lib.ppo_episode_stats_ints_hist_offset()
        reps = _lib_const("PPO_STATS_INTS_REPLICAS")       # partial histograms; their sum mod 2^32 is this rank's window
        parts = self._stats_state[off:off + reps * (RB + LB) * 4].view(torch.int32).view(reps, RB + LB)
        h = parts.to(torch.int64).sum(0) & 0xFFFFFFFF      # this rank's counts (int64: the job's sum cannot wrap)
        h = ppo_dist.sum_tensor(h, "ep_hist_ints").cpu().numpy()
        n = int(h[:RB].sum())
        if n == 0:                                         # no episode has finished anywhere yet
            return job_reward
        if job_bad != 0:                                   # a value that is not a small integer somewhere in the job's window
            self._note_stats([float("nan")] * 5)
            return job_reward
        win_max, p5, p50, p95, med_len = _direct_select(h[:RB], h[RB:], R0)
        self._note_stats([job_max if job_max > -float("inf") else win_max, p5, p50, p95, med_len])
        if not (self._stats_host["max"] >= win_max):
            self._stats_host["max"] = np.float32(win_max)
        return job_reward

    def _job_wide_host_stats(self) -> float:
        """job_wide_episode_stats for host-stepped envs (BASELINE configs[4]: Ant sharded over the node): every rank's
        deques in one padded gather, then the reference's own percentiles over all of them."""
        N = self.num_envs
        cap = N * 100
        has = self.envs is not None and hasattr(self.envs[0], "return_queue")
        rets = np.concatenate([np.array(e.return_queue, dtype=np.float64).reshape(-1) for e in self.envs]) if has else np.zeros(0)
        lens = np.concatenate([np.array(e.length_queue, dtype=np.float64).reshape(-1) for e in self.envs]) if has else np.zeros(0)
        mine = np.zeros(3 + 2 * cap, dtype=np.float64)
        mine[0], mine[1], mine[2] = float(self.total_return), rets.size, lens.size
        mine[3:3 + rets.size] = rets
        mine[3 + cap:3 + cap + lens.size] = lens
        rows = ppo_dist.gather_rows(torch.from_numpy(mine), "ep_deques_host").numpy()
        all_r = np.concatenate([r[3:3 + int(r[1])] for r in rows])
        all_l = np.concatenate([r[3 + cap:3 + cap + int(r[2])] for r in rows])
        self._local_max = np.float32(rets.max()) if rets.size else getattr(self, "_local_max", np.float32(np.nan))
        self._episode_percentiles(torch.from_numpy(all_r), torch.from_numpy(all_l))
        return float(rows[:, 0].sum())

    max_ep_return = property(lambda self: self._stat("max"),
                             lambda self, v: self._stats_host.__setitem__("max", v))
    lower_ep_return = property(lambda self: self._stat("lower"),
                               lambda self, v: self._stats_host.__setitem__("lower", v))
    median_ep_return = property(lambda self: self._stat("median"),
                                lambda self, v: self._stats_host.__setitem__("median", v))
    upper_ep_return = property(lambda self: self._stat("upper"),
                               lambda self, v: self._stats_host.__setitem__("upper", v))
    median_ep_length = property(lambda self: self._stat("length"),
                                lambda self, v: self._stats_host.__setitem__("length", v))

    # -- checkpoint (no counterpart in the reference, which keeps no state_dict) -------
    def state_dict(self) -> dict:
        """Everything a resumed run needs to continue the env batch bit for bit: physics state,
        TimeLimit / episode counters (the Philox reset position), statistics rings, the carried
        observations and the rollout counter (the Philox sampling position)."""
        if not self.native:
            raise NotImplementedError("host-stepped gym envs are opaque objects; only device-resident env "
                                      "batches can be checkpointed")
        b = self.batch
        torch.cuda.synchronize()
        _ = self.max_ep_return                       # drain the statistics stream into _stats_host
        sd = {name: getattr(b, name).cpu() for name in _BATCH_TENSORS}
        sd.update({"kind": b.kind, "num_envs": b.num_envs, "seed": b.seed, "env_base": b.env_base, "ring": b.ring,
                   "t_states": self.t_states.cpu(), "rollouts": self._rollouts, "global_steps": self.global_steps,
                   "stats_host": {k: float(v) for k, v in self._stats_host.items()},
                   "stats5": None if self._stats5 is None else self._stats5.cpu()})
        return sd

    def load_state_dict(self, sd: dict):
        if not self.native:
            raise NotImplementedError("only device-resident env batches can be restored")
        b = self.batch
        for k in ("kind", "num_envs", "ring"):
            if sd[k] != getattr(b, k):
                raise ValueError(f"checkpoint env batch has {k}={sd[k]}, this run has {getattr(b, k)}")
        for name in _BATCH_TENSORS:
            getattr(b, name).copy_(sd[name])
        self.invalidate_episode_stats_state()
        b.seed, b.env_base = int(sd["seed"]), int(sd["env_base"])
        self.t_states = sd["t_states"].to(self.device)
        self._rollouts, self.global_steps = int(sd["rollouts"]), int(sd["global_steps"])
        self._stats_host = {k: np.float32(v) if k == "max" else v for k, v in sd["stats_host"].items()}
        self._stats_dev = None
        self._stats_pending = False
        if sd["stats5"] is not None:                 # the running max lives in this device buffer
            if self._stats5 is None:
                self._device_episode_stats()         # allocates the buffers
            self._stats5.copy_(sd["stats5"])
            self._stats_dev = None
        torch.cuda.synchronize()

    def _rollout_host(self):
        """Host-stepped gym envs; the control flow of reference src/ppo.py:320-398 with the
        per-actor bookkeeping done on whole [N] arrays."""
        T, N, dev = self.steps_per_env, self.num_envs, self.device
        self._alloc(dev)
        end_states = [[] for _ in range(N)]
        total = np.float32(0.0)
        for t in range(T):
            a, lp, v, _ = self.agent.get_actions_and_values(self.t_states, actions=None)
            new_states, rewards, dones, truncs = self.vec_step(a)
            d_np, tr_np = dones.cpu().numpy(), truncs.cpu().numpy().copy()
            for r in rewards.cpu().numpy():
                total = np.float32(total + r)          # reference :348 accumulates in np.float32
            new_states = new_states.clone()
            for n in range(N):
                ended = d_np[n] + tr_np[n]
                can_reset = True
                if ended == 0 and t == T - 1:          # reference :353-356
                    ended, can_reset = 1, False
                    tr_np[n] += 1
                if ended > 0:                          # reference :377-382
                    end_states[n].append(new_states[n].cpu().clone())
                    if can_reset:
                        new_states[n] = self.env_reset(n)
            self.global_steps += N
            self.states[t], self.actions[t], self.rewards[t] = self.t_states, a, rewards
            self.done_flags[t] = dones
            self.trunc_flags[t] = torch.from_numpy(tr_np).to(dev)
            self.values[t], self.log_probs[t] = v, lp
            self.t_states = new_states
        self.total_return = total
        self.end_states = torch.cat([torch.stack(e, dim=0) for e in end_states], dim=0).to(dev)
        self._end_n = self._end_t = None
        self._ev_ready = False
        if not hasattr(self.envs[0], "return_queue"):  # reference :404-405
            return
        rets = np.concatenate([np.array(e.return_queue, dtype=np.float64).reshape(-1) for e in self.envs])
        lens = np.concatenate([np.array(e.length_queue, dtype=np.float64).reshape(-1) for e in self.envs])
        self._episode_percentiles(torch.from_numpy(rets), torch.from_numpy(lens))

    def _episode_percentiles(self, rets: torch.Tensor, lens: torch.Tensor):
        """reference src/ppo.py:407-417: running max, 5/50/95 percentiles (linear), median length"""
        if rets.numel() > 0:
            srt = torch.sort(rets.to(torch.float64)).values
            q = _percentiles(srt, (5.0, 50.0, 95.0)).cpu().numpy()
            mx = np.float32(srt[-1].item())
            if np.isnan(self.max_ep_return) or mx > self.max_ep_return:
                self.max_ep_return = mx
            self.lower_ep_return, self.median_ep_return, self.upper_ep_return = (np.float64(x) for x in q)
        if lens.numel() > 0:
            srt = torch.sort(lens.to(torch.float64)).values
            self.median_ep_length = np.float64(_percentiles(srt, (50.0,)).item())


_LIB_CONSTS = {"PPO_STATS_INTS_RET_ZERO": 512, "PPO_STATS_INTS_RET_BINS": 1280, "PPO_STATS_INTS_LEN_BINS": 768,
               "PPO_STATS_INTS_LEN_MAX": 512, "PPO_STATS_INTS_REPLICAS": 8}      # include/ppo_amd.h (tests/test_abi.py checks them against the header)


def _lib_const(name: str) -> int:
    return _LIB_CONSTS[name]


def _select_ranks(n: int):
    """Ranks (0-based, among n values) of the nine order statistics behind the reference's statistics: lower / upper
    neighbour of the 5th, 50th and 95th percentile of the returns (np.percentile, method "linear"), their maximum,
    lower / upper neighbour of the median length -- csrc/stats.hip sel_target_rank."""
    def pair(q):
        lo = int(np.floor(q * float(n - 1)))
        return lo, min(lo + 1, n - 1)
    p5, p50, p95 = pair(0.05), pair(0.50), pair(0.95)
    return [p5[0], p5[1], p50[0], p50[1], p95[0], p95[1], n - 1, p50[0], p50[1]]


def _bucket_of(cum: np.ndarray, rank: int):
    """(bucket holding the entry of 0-based rank `rank`, its rank inside that bucket) from inclusive counts."""
    bkt = int(np.searchsorted(cum, rank, side="right"))
    return bkt, rank - (int(cum[bkt - 1]) if bkt > 0 else 0)


def _direct_select(h_ret: np.ndarray, h_len: np.ndarray, ret_zero: int):
    """[window max, p5, p50, p95 of the returns, median length] from DIRECT histograms (csrc/stats.hip k_sel_incr_ints:
    h_ret[ret_zero + r] counts returns equal to the integer r, h_len[l] lengths equal to l): the reference's np.max /
    np.percentile (src/ppo.py:407-417) read off two prefix sums."""
    n = int(h_ret.sum())
    ranks = _select_ranks(n)
    cum = [np.cumsum(h_ret), np.cumsum(h_len)]
    v = [float(_bucket_of(cum[0], r)[0] - ret_zero) if t < 7 else float(_bucket_of(cum[1], r)[0]) for t, r in enumerate(ranks)]
    return [v[6], _lerp(v[0], v[1], n, 0.05), _lerp(v[2], v[3], n, 0.50), _lerp(v[4], v[5], n, 0.95), _lerp(v[7], v[8], n, 0.50)]


def _key_to_float(keys: np.ndarray) -> np.ndarray:
    """Inverse of the order-preserving f32 -> u32 map of csrc/stats.hip (f2key), as float64 values."""
    k = keys.astype(np.uint32)
    bits = np.where(k & np.uint32(0x80000000), k & np.uint32(0x7FFFFFFF), ~k)
    return bits.astype(np.uint32).view(np.float32).astype(np.float64)


def _lerp(a: float, b: float, n: int, q: float) -> float:
    pos = q * float(n - 1)
    return float(a + (b - a) * (pos - np.floor(pos)))


def _percentiles(sorted_vals: torch.Tensor, qs) -> torch.Tensor:
    """np.percentile(..., method='linear') on an ascending 1-d tensor."""
    n = sorted_vals.numel()
    pos = torch.tensor(qs, dtype=torch.float64, device=sorted_vals.device) / 100.0 * (n - 1)
    lo = pos.floor().long()
    hi = torch.clamp(lo + 1, max=n - 1)
    frac = pos - lo.to(torch.float64)
    return sorted_vals[lo] + (sorted_vals[hi] - sorted_vals[lo]) * frac


# ----------------------------------------------------------------------------------------
class PPO:
    """PPO trainer with the reference's constructor, methods and metrics (src/ppo.py:420-763)."""

    def __init__(self, discount_factor, gae_factor, norm_adv, clip_va_loss, conv_net, joint_network,
                 use_gpu, **kwargs):
        self.discount_factor, self.gae_factor = discount_factor, gae_factor
        self.norm_adv, self.clip_va_loss = norm_adv, clip_va_loss
        self.conv_net, self.joint_network, self.use_gpu = conv_net, joint_network, use_gpu
        self.project_name = kwargs.get("project_name", None)
        self.seed = kwargs.get("seed", None)
        self.agent: Agent = None
        self.vec_env: SyncVecEnv = None
        self.pi_optimizer = None
        self.va_optimizer = None
        # where the MLP and the rollout buffers live; the kernels always run on the HIP device
        self.device = "cuda" if (use_gpu and torch.cuda.is_available()) else "cpu"
        self.updates = 0
        self._lr_anneal = 1.0
        self.last_metrics: dict = {}
        # Discrete heads: log-prob, entropy, ratio and losses in one kernel (ppo_loss_logits_fwd_bwd);
        # False keeps torch.distributions.Categorical in front of ppo_loss_fwd_bwd.
        self.fused_discrete_loss = True
        # dense Discrete agents (CartPole / Acrobot): the whole minibatch step in one kernel; PPO_AMD_FUSED_UPDATE=0
        # keeps the PyTorch MLP around the loss kernel
        self.fused_dense_update = os.environ.get("PPO_AMD_FUSED_UPDATE", "1") != "0"
        # with it: clip_grad_norm_ + Adam.step of a minibatch as one launch (ppo_clip_adam_step) instead of PyTorch's 17
        self.fused_optimizer = os.environ.get("PPO_AMD_FUSED_OPTIMIZER", "1") != "0"
        self.fused_epoch_stats = os.environ.get("PPO_AMD_EPOCH_STATS", "1") != "0"   # ops.adv_stats_epoch, see train_step
        self._last_perm_keys = None
        # the epoch's shuffled index list from ppo_shuffle_indices (one pass; torch.randperm's 64-bit key sort costs
        # more than a minibatch step at C2 scale); PPO_AMD_SHUFFLE=torch keeps torch.randperm
        self.device_shuffle = os.environ.get("PPO_AMD_SHUFFLE", "kernel") != "torch"
        self._shuffle = None         # [seed, epochs shuffled so far]: the permutation is a function of the two
        self.minibatch_hook = None   # tests: called with a dict per minibatch (indices, losses) when set
        self._idle = False           # True while nothing of ours is queued on the device (set by a host read-back)
        self._param_list = None      # list(agent.parameters()), built on first use

    # -- hot path: GAE --------------------------------------------------------------------
    def compute_advantages(self, rewards, values, end_values, done_flags, trunc_flags):
        """reference src/ppo.py:452-488 -> [T, N] f32 on self.device (ppo_gae_ragged)."""
        adv = ops.gae_ragged(rewards, values, end_values, done_flags, trunc_flags,
                             self.discount_factor, self.gae_factor)
        return adv.to(self.device)

    # -- hot path: loss -------------------------------------------------------------------
    def compute_losses(self, prob_ratios, curr_values, prev_values, advantages, clip_ratio):
        """reference src/ppo.py:490-527 -> (policy_loss, value_loss, clip_frac, kl_div), 0-d f32;
        the first two carry gradients to prob_ratios / curr_values (ppo_loss_fwd_bwd)."""
        return ops.ppo_losses(prob_ratios, curr_values, None, prev_values, advantages, clip_ratio,
                              self.norm_adv, self.clip_va_loss, ratio_input=True)

    def _epoch_permutation(self, n: int, dev) -> torch.Tensor:
        """reference src/ppo.py:575-576 (np.random.shuffle(indices)): the rows of this rank's rollout in random order."""
        self._last_perm_keys = None
        if dev.type != "cuda" or not self.device_shuffle:
            return torch.randperm(n, device=dev)
        if self._shuffle is None:
            # an explicit seed, or one drawn from torch's CPU generator (so torch.manual_seed still governs the run)
            seed = int(self.seed) if self.seed is not None else int(torch.randint(0, 2**62, (1,)).item())
            self._shuffle = [seed, 0]
        stream = (ppo_dist.rank() << 40) + self._shuffle[1]          # every rank shuffles its own shard
        self._shuffle[1] += 1
        self._last_perm_keys = (self._shuffle[0], stream)              # what ops.adv_stats_epoch needs to follow it
        return ops.shuffle_indices(n, self._shuffle[0], stream)

    # -- one update -----------------------------------------------------------------------
    def train_step(self, num_epochs, batch_size, critic_coef, entropy_coef, clip_ratio, target_div,
                   max_grad_norm, lr_anneal):
        """reference src/ppo.py:529-673"""
        ve = self.vec_env
        num_envs, steps_per_env = ve.num_envs, ve.steps_per_env
        world = ppo_dist.world_size()
        # Data parallel: batch_size is the GLOBAL minibatch, every rank contributes batch_size / world
        # samples of its own env shard, so the number of optimiser steps and the samples behind each
        # gradient are those of the single-process run (reference src/ppo.py:578-588).
        if batch_size % world != 0 or batch_size < world:
            raise ValueError(f"batch_size={batch_size} must be a positive multiple of world_size={world}")
        local_batch = batch_size // world

        if not self._idle:                         # the previous update's kernels are not part of this bracket
            _sync()                                # (skipped when that update ended in a host read-back: nothing is queued)
        self._idle = False
        rollout_start = time.time()
        ve.rollout()
        with torch.no_grad():
            advantages = self._rollout_advantages(ve)
        if ve.native:
            ve._flush_stats()                      # the reference's rollout() ends with the episode statistics (:404-417)
        _sync_spin()
        rollout_time = ppo_dist.max_scalar(time.time() - rollout_start)   # the job waits for its slowest rank
        env_steps_per_sec = steps_per_env * num_envs * world / rollout_time

        updates_start = time.time()
        data_size = steps_per_env * num_envs
        # the first epoch's shuffle goes to the (idle) device before the host-side bookkeeping below
        first_perm = self._epoch_permutation(data_size, ve.states.device)
        states = ve.states.detach().flatten(0, 1)  # no gradient wrt observations is ever used
        actions = ve.actions.flatten(0, 1)
        log_probs = ve.log_probs.flatten(0, 1)
        values = ve.values.flatten(0, 1)
        advantages = advantages.flatten(0, 1)
        dev = states.device

        early = torch.zeros((), dtype=torch.bool, device=dev) if target_div is not None else None
        clip_terms = []                                        # clip_frac of every minibatch, summed once at the end
        grad_steps_done = 0
        if self._param_list is None or self._param_list[0] is not self.agent:
            self._param_list = (self.agent, [p for p in self.agent.parameters()])
        params = self._param_list[1]
        fused_head = (self.fused_discrete_loss and self.agent.discrete and self.agent.action_space.n <= 8
                      and actions.dtype == torch.int32)
        fused_gather = (dev.type == "cuda" and actions.dtype in (torch.int32, torch.float32)
                        and states.dtype == torch.float32)
        # dense 64-64 tanh Agent + Discrete head: forward, loss, backward and weight gradients of a minibatch in
        # one kernel (ppo_dense_update_grads) instead of the PyTorch MLP around the loss kernel
        fused_update = (self.fused_dense_update and fused_head and fused_gather and ops.dense_update_supported(self.agent))
        # its 64 x 64 products in the arithmetic the rollout just ran with: the split kernels inside the limits
        # checked when the parameters were packed (ops.split_ok), exact f32 products otherwise
        upd_mode = ops.NET_F16_SPLIT if ve.net_mode_used == ops.NET_F16_SPLIT else ops.NET_F32
        fused_opt = (fused_update and self.fused_optimizer and ops.clip_adam_supported(self.pi_optimizer)
                     and ops.clip_adam_supported(self.va_optimizer))
        self.last_paths = {"fused_head": bool(fused_head), "fused_gather": bool(fused_gather),
                           "fused_update": bool(fused_update), "fused_optimizer": bool(fused_opt)}
        # one row of {policy_loss, value_loss, clip_frac, kl_div, entropy} per minibatch, written by the update kernel
        out_rows = (torch.empty((num_epochs * ((data_size + local_batch - 1) // local_batch), 5), dtype=torch.float32,
                                device=dev) if fused_update else None)
        for _epoch in range(num_epochs):
            perm = first_perm if _epoch == 0 else self._epoch_permutation(data_size, dev)   # reference :575-576 (np.random.shuffle)
            # the advantage statistics of all of the epoch's minibatches from one streaming pass (the device
            # shuffle's bijection run backwards) instead of one gather per minibatch
            epoch_stats = None
            if (fused_update and self.norm_adv and self._last_perm_keys is not None and self.fused_epoch_stats
                    and (data_size + local_batch - 1) // local_batch <= ops.ADV_STATS_EPOCH_MAX_MB):
                epoch_stats = ops.adv_stats_epoch(advantages, *self._last_perm_keys, local_batch)
                if world > 1:                                  # minibatch normalisation stays global: one collective per epoch
                    epoch_stats = ppo_dist.merged_adv_stats_epoch(epoch_stats)
            for start in range(0, data_size, local_batch):
                idx = perm[start:start + local_batch]          # last minibatch may be short (:581)
                if fused_update:                               # the update kernel gathers as it goes
                    mb_adv = None
                elif fused_gather:                             # one launch for the five gathers (:582-588)
                    mb_states, mb_actions, mb_log_probs, mb_values, mb_adv = ops.minibatch_gather(
                        idx, states, actions, log_probs, values, advantages)
                else:
                    mb_states, mb_actions = states[idx], actions[idx]
                    mb_log_probs, mb_values, mb_adv = log_probs[idx], values[idx], advantages[idx]
                stats3 = None
                local3 = epoch_stats[start // local_batch] if epoch_stats is not None else None
                if local3 is not None:
                    stats3 = local3                            # (already merged over ranks)
                elif self.norm_adv and world > 1:              # keep minibatch normalisation global
                    stats3 = ppo_dist.merged_adv_stats(
                        ops.adv_stats(advantages, idx) if fused_update else ops.adv_stats(mb_adv))
                if fused_update:
                    if self.norm_adv and stats3 is None:
                        stats3 = ops.adv_stats(advantages, idx)
                    out5, groups = ops.dense_update_grads(
                        self.agent, states, actions, log_probs, values, advantages, clip_ratio, self.norm_adv,
                        self.clip_va_loss, c_pi=lr_anneal, c_v=critic_coef * lr_anneal, c_ent=entropy_coef * lr_anneal,
                        stats3=stats3, idx=idx, net_mode=upd_mode, out5=out_rows[grad_steps_done],
                        reuse_bucket=self.minibatch_hook is None)
                    policy_loss, critic_loss, clip_frac, kl_div, entropy = out5.unbind(0)
                elif fused_head:   # Categorical log-prob + entropy + ratio + losses in one pass (:186-216, :595-599)
                    logits, new_values = self.agent.policy_and_values(mb_states)
                    policy_loss, critic_loss, clip_frac, kl_div, entropy = ops.ppo_losses_discrete(
                        logits, mb_actions, new_values, mb_log_probs, mb_values, mb_adv, clip_ratio,
                        self.norm_adv, self.clip_va_loss, stats3=stats3)
                else:
                    _, new_log_probs, new_values, entropy = self.agent.get_actions_and_values(mb_states, mb_actions)
                    policy_loss, critic_loss, clip_frac, kl_div = ops.ppo_losses(
                        new_log_probs, new_values, mb_log_probs, mb_values, mb_adv, clip_ratio,
                        self.norm_adv, self.clip_va_loss, ratio_input=False, stats3=stats3)  # :595-599
                if fused_update:                               # gradients are already there: :602-625 without backward()
                    for k, (opt, grp) in enumerate(zip((self.pi_optimizer, self.va_optimizer), groups)):
                        if not fused_opt:                       # Adam.step skips parameters without a gradient (:614-625)
                            opt.zero_grad()
                        for prm, grad in grp:
                            if prm.grad is not grad:
                                prm.grad = grad
                        ppo_dist.allreduce_flat(grp.flat)       # the group's gradients are views of one bucket
                        if fused_opt:                           # :608-609 as one launch on that bucket
                            ops.clip_adam_step(opt, grp, max_grad_norm)
                        else:
                            nn.utils.clip_grad_norm_(params, max_grad_norm)
                            opt.step()
                    if not self.joint_network and self.minibatch_hook is not None:
                        critic_loss = critic_coef * critic_loss    # (:619; without a hook the logged value is scaled on the host)
                elif self.joint_network:                       # reference :602-609
                    total = (policy_loss + critic_coef * critic_loss - entropy_coef * entropy) * lr_anneal
                    self.pi_optimizer.zero_grad()
                    total.backward()
                    ppo_dist.allreduce_grads(params)
                    nn.utils.clip_grad_norm_(params, max_grad_norm)
                    self.pi_optimizer.step()
                else:                                          # reference :610-625
                    full_policy = (policy_loss - entropy_coef * entropy) * lr_anneal
                    self.pi_optimizer.zero_grad()
                    full_policy.backward(retain_graph=True)
                    ppo_dist.allreduce_grads(params)
                    nn.utils.clip_grad_norm_(params, max_grad_norm)
                    self.pi_optimizer.step()
                    critic_loss = critic_coef * critic_loss
                    self.va_optimizer.zero_grad()
                    (critic_loss * lr_anneal).backward()
                    ppo_dist.allreduce_grads(params)
                    nn.utils.clip_grad_norm_(params, max_grad_norm)
                    self.va_optimizer.step()
                if target_div is not None:                     # reference :628-629
                    kl_all = ppo_dist.mean_scalar(kl_div.detach())
                    early = early | (kl_all > target_div).to(dev)
                clip_terms.append(clip_frac.detach())
                grad_steps_done += 1
                if self.minibatch_hook is not None:
                    self.minibatch_hook(dict(update=self.updates, epoch=_epoch, start=start, idx=idx,
                                             policy_loss=policy_loss.detach(), critic_loss=critic_loss.detach(),
                                             entropy=entropy.detach(), kl_div=kl_div.detach()))
            if target_div is not None and bool(early.item()):  # checked at the end of an epoch (:634)
                break
        # the next rollout's parameters are packed here, right behind the last optimizer.step(): the
        # rollout bracket of the next update starts with a kernel launch, not with host-side packing
        diag = ve.prepack() if (ve.native and ve.fused_rollout and dev.type == "cuda") else None
        # one device->host copy for every scalar of this update (it is also the update's final sync)
        # -- together with the rollout's episode statistics and reward sum when those still live on the device
        rows_form = fused_update and grad_steps_done > 0
        if rows_form:       # the update kernel's table: the last minibatch's row as it is + the sum of the clip_frac column
            pieces = [out_rows[grad_steps_done - 1], out_rows[:grad_steps_done, 2].sum(dtype=torch.float32).reshape(1)]
        else:
            clip_sum = torch.stack(clip_terms).sum(dtype=torch.float32).to(dev)
            scalars = [policy_loss.detach().to(dev), critic_loss.detach().to(dev), entropy.detach().to(dev),
                       kl_div.detach().to(dev), clip_sum]
            pieces = [torch.stack([s.to(torch.float32).reshape(()) for s in scalars])]
        n_own = 6 if rows_form else 5
        if diag is not None:
            pieces.append(diag.to(dev))
        job_reward = ve.job_wide_episode_stats() if world > 1 else None      # every rank, once per update (collectives)
        ve_vals, ve_take = ve.pending_device_scalars() if (ve.native and dev.type == "cuda") else (None, None)
        if ve_vals is not None and ve_vals.device == dev:
            pieces.append(ve_vals)                             # f64: torch.cat promotes the f32 pieces exactly
        else:
            ve_take = None
        host = (pieces[0] if len(pieces) == 1 else torch.cat(pieces)).cpu().tolist()
        if diag is not None:
            ve.note_split_diag(host[n_own:n_own + 3])
        if ve_take is not None:
            ve_take(host[n_own + (3 if diag is not None else 0):])
        if rows_form:       # {policy_loss, value_loss, clip_frac, kl_div, entropy} + clip sum -> the order used below
            scale = 1.0 if self.joint_network else critic_coef          # :619 (the table holds the unscaled value loss)
            host = [host[0], host[1] * scale, host[4], host[3], host[5]]
        updates_time = ppo_dist.max_scalar(time.time() - updates_start)
        self._idle = dev.type == "cuda" and world == 1
        grad_steps_per_sec = grad_steps_done / updates_time
        global_steps_per_sec = steps_per_env * num_envs * world / (rollout_time + updates_time)

        policy_loss, critic_loss, entropy, kl_div = host[0], host[1], host[2], host[3]
        clip_frac = float(host[4] / max(grad_steps_done, 1))
        rollout_return = float(ve.total_return) / num_envs if world == 1 else job_reward / (num_envs * world)   # :318, :348 over all envs
        ep_return_max = float(ve.max_ep_return)
        self.last_metrics = {  # reference :656-670, same names
            "utils/ep_return_0.50": float(ve.median_ep_return),
            "utils/ep_return_0.05": float(ve.lower_ep_return),
            "utils/ep_return_0.95": float(ve.upper_ep_return),
            "utils/env_steps_per_sec": env_steps_per_sec,
            "utils/grad_steps_per_sec": grad_steps_per_sec,
            "utils/global_steps_per_sec": global_steps_per_sec,
            "losses/policy_loss": policy_loss,
            "losses/critic_loss": critic_loss,
            "losses/entropy": entropy,
            "metrics/clip_frac": clip_frac,
            "metrics/kl_div": kl_div,
            "metrics/roll_return": rollout_return,
            "metrics/ep_return_max": ep_return_max,
            "metrics/ep_length_0.50": float(ve.median_ep_length),
        }
        if ve.native and ve.net_mode_used is not None:     # (an addition: which network arithmetic this update's rollout ran)
            self.last_metrics["utils/net_mode_used"] = ops.NET_NAMES[ve.net_mode_used]
        self._log(self.last_metrics, step=self.updates * steps_per_env * num_envs * world)
        return policy_loss, critic_loss, entropy, clip_frac, kl_div, rollout_return, ep_return_max

    def _rollout_advantages(self, ve: SyncVecEnv) -> torch.Tensor:
        """Bootstrap values + GAE for the rollout just collected (reference src/ppo.py:554-557)."""
        if getattr(ve, "_ev_ready", False):   # whole-rollout kernel already wrote V(terminal obs) where trunc != 0
            return ops.gae_dense_packed(ve.rewards, ve.values, ve._ev_dense, ve._flag_words,
                                        self.discount_factor, self.gae_factor)
        end_values = self.agent.get_values(ve.end_states)
        return self._advantages(ve, end_values)

    def _scatter_end_values(self, ve: SyncVecEnv, end_values: torch.Tensor) -> torch.Tensor:
        """Dense bootstrap buffer: V(terminal obs) written at the (t, n) that raised the flag.  Only
        those cells are ever read by ppo_gae_dense, so the buffer is reused without clearing."""
        if ve._ev_dense is None or ve._ev_dense.shape != ve.rewards.shape:
            ve._ev_dense = torch.empty_like(ve.rewards)
        ve._ev_dense[ve._end_t, ve._end_n] = end_values
        return ve._ev_dense

    def _advantages(self, ve: SyncVecEnv, end_values: torch.Tensor) -> torch.Tensor:
        if ve.native and ve._end_t is not None and ve.rewards.device.type == "cuda":
            ev_dense = self._scatter_end_values(ve, end_values)
            return ops.gae_dense(ve.rewards, ve.values, ev_dense, ve.done_flags, ve.trunc_flags,
                                 self.discount_factor, self.gae_factor)
        return self.compute_advantages(ve.rewards, ve.values, end_values, ve.done_flags, ve.trunc_flags)

    def _log(self, metrics: dict, step: int):
        path = os.environ.get("PPO_AMD_METRICS")
        if path and ppo_dist.rank() == 0:
            with open(path, "a") as f:
                f.write(json.dumps({"step": step, **metrics}) + "\n")

    # -- checkpoint / resume (absent in the reference) ------------------------------
    def state_dict(self) -> dict:
        return {"agent": self.agent.state_dict(), "pi_optimizer": self.pi_optimizer.state_dict(),
                "va_optimizer": self.va_optimizer.state_dict(), "vec_env": self.vec_env.state_dict(),
                "updates": self.updates, "lr_anneal": self._lr_anneal,
                "shuffle": list(self._shuffle) if self._shuffle is not None else None,
                "rng_cpu": torch.get_rng_state(),
                "rng_cuda": torch.cuda.get_rng_state() if torch.cuda.is_available() else None}

    def load_state_dict(self, sd: dict):
        self.agent.load_state_dict(sd["agent"])
        self.pi_optimizer.load_state_dict(sd["pi_optimizer"])
        self.va_optimizer.load_state_dict(sd["va_optimizer"])
        self.vec_env.load_state_dict(sd["vec_env"])
        self.updates, self._lr_anneal = int(sd["updates"]), float(sd["lr_anneal"])
        shuffle = sd.get("shuffle")
        self._shuffle = [int(shuffle[0]), int(shuffle[1])] if shuffle is not None else None
        torch.set_rng_state(sd["rng_cpu"])
        if sd["rng_cuda"] is not None and torch.cuda.is_available():
            torch.cuda.set_rng_state(sd["rng_cuda"])

    def save_checkpoint(self, path: str):
        """One file per rank under data parallelism (each rank owns its env shard)."""
        tmp = f"{path}.tmp"
        torch.save(self.state_dict(), tmp)
        os.replace(tmp, path)

    def load_checkpoint(self, path: str):
        self.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))

    # -- training loop ----------------------------------------------------------------------
    def train(self, env_fn, num_updates, num_envs, steps_per_env, num_epochs, batch_size, critic_coef,
              entropy_coef, clip_ratio, max_grad_norm, learning_rate, target_div=None, render_every=0,
              render_fps=0.0, early_stop_reward=None, checkpoint_path=None, checkpoint_every=0,
              resume_from=None):
        """reference src/ppo.py:675-763.  Under torch.distributed (one process per GPU) `num_envs`
        is the GLOBAL env count; every rank owns num_envs / world_size of them.
        checkpoint_path / checkpoint_every / resume_from are additions (the reference cannot resume):
        a resumed run continues bit for bit where the saved one stopped (rank r uses `<path>.rank<r>`
        when world_size > 1)."""
        test_env = env_fn()
        test_obs_shape = np.asarray(test_env.reset()[0]).shape
        state_space = test_env.observation_space
        if hasattr(state_space, "_shape") and test_obs_shape != state_space.shape:
            state_space._shape = test_obs_shape            # reference :722-723
        action_space = test_env.action_space
        del test_env

        world, rank = ppo_dist.world_size(), ppo_dist.rank()
        if num_envs % world != 0:
            raise ValueError(f"num_envs={num_envs} must be divisible by world_size={world}")
        local_envs = num_envs // world
        self.agent = Agent(state_space, action_space, conv_net=self.conv_net, joint_net=self.joint_network,
                           device=self.device)
        ppo_dist.broadcast_params(self.agent)
        self.vec_env = SyncVecEnv(env_fn, local_envs, steps_per_env, render_every, render_fps, self.agent,
                                  seed=self.seed, env_base=rank * local_envs)
        # both optimisers cover ALL parameters, as in the reference (:732-733)
        self.pi_optimizer = optim.Adam(self.agent.parameters(), lr=learning_rate, eps=1e-5)
        self.va_optimizer = optim.Adam(self.agent.parameters(), lr=learning_rate, eps=1e-5)
        self._lr_anneal = 1.0
        suffix = f".rank{rank}" if world > 1 else ""
        if resume_from:
            self.load_checkpoint(resume_from + suffix)

        it = range(self.updates, num_updates)
        pbar = (_tqdm.trange(self.updates, num_updates, leave=True)
                if (_tqdm is not None and rank == 0) else None)
        for _update in (pbar if pbar is not None else it):
            pi_loss, va_loss, entropy, clip_frac, kl_div, roll_return, ep_return_max = self.train_step(
                num_epochs, batch_size, critic_coef, entropy_coef, clip_ratio, target_div, max_grad_norm,
                self._lr_anneal)
            if num_updates > 1:                            # the reference divides by zero here for 1 update (:748)
                self._lr_anneal -= 0.999 / (num_updates - 1)
            self.updates += 1
            if checkpoint_path and checkpoint_every > 0 and self.updates % checkpoint_every == 0:
                self.save_checkpoint(checkpoint_path + suffix)
            if pbar is not None:
                pbar.set_postfix({"pi_loss": f"{pi_loss:.3f}", "va_loss": f"{va_loss:.3f}",
                                  "entropy": f"{entropy:.3f}", "return": f"{roll_return:.3f}",
                                  "max_ep_ret": f"{ep_return_max:.3f}", "kl_div": f"{kl_div:.4f}",
                                  "clip_frac": f"{clip_frac:.3f}"}, refresh=False)
                # (refresh=False: the bar redraws at tqdm's own rate limit, 10 Hz, instead of once per update --
                # a forced redraw is ~0.1 ms of host time between two updates while the device waits)
            # reference :758.  Data parallel: the decision is taken by ALL ranks together (MAX all-reduce) -- a rank
            # that left the loop alone would leave the others waiting in the next update's first collective.
            if early_stop_reward is not None and ppo_dist.any_rank(ep_return_max >= early_stop_reward, "early_stop_reward"):
                if pbar is not None:
                    pbar.close()
                if rank == 0:
                    print("Early stop reward reached.")
                break


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


_spin_events = {}          # one per device: an event belongs to the device it was first recorded on


def _sync_spin():
    """_sync() for the end of the rollout bracket: poll an event recorded behind the bracket's last kernel instead of
    blocking in hipDeviceSynchronize -- the blocked thread is woken ~20 us after the kernel has finished, the poll sees it
    within a few (the bracket is ~0.5 ms and the host has nothing else to do: src/ppo.py:544-560 blocks there too).
    PPO_AMD_SPIN_SYNC=0: block."""
    if not torch.cuda.is_available():
        return
    if os.environ.get("PPO_AMD_SPIN_SYNC", "1") == "0":
        torch.cuda.synchronize()
        return
    d = torch.cuda.current_device()
    ev = _spin_events.get(d)
    if ev is None:
        ev = _spin_events[d] = torch.cuda.Event(enable_timing=False, blocking=False)
    ev.record()                                    # current stream: everything the bracket launched is on it
    while not ev.query():
        pass
