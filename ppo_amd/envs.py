"""Device-resident CartPole-v1 / Acrobot-v1 batches stepped by the gfx950 kernels.

`DeviceEnvBatch` owns the per-env state in HBM (fp64 physics state, TimeLimit counters,
episode-statistics accumulators and rings, Philox reset counters) and calls
ppo_env_reset / ppo_env_step (include/ppo_amd.h).  `DeviceEnv` is the one-env view with the
gym.Env protocol the reference's SyncVecEnv and PPO.train probe (reference
src/ppo.py:272,296,718-724): it is what `env_cartpole.train_fn()` returns here in place of
`gym.make("CartPole-v1")` + RecordEpisodeStatistics (reference src/env_cartpole.py:6-9).
There is no host physics anywhere in this package.
"""
from __future__ import annotations

import math
import os
from collections import deque

import numpy as np
import torch

from . import _lib, ops, spaces

KINDS = {"cartpole": 0, "acrobot": 1}
OBS_DIM = {0: 4, 1: 6}
NUM_ACTIONS = {0: 2, 1: 3}
STAT_RING = 100  # RecordEpisodeStatistics(deque_size=100), reference src/env_cartpole.py:8


def default_seed() -> int:
    s = os.environ.get("PPO_AMD_SEED")
    return int(s) if s is not None else int.from_bytes(os.urandom(8), "little")


def make_spaces(kind: int):
    if kind == 0:
        high = np.array([4.8, np.inf, 24 * math.pi / 360 * 2, np.inf], dtype=np.float32)
    else:
        high = np.array([1.0, 1.0, 1.0, 1.0, 4 * math.pi, 9 * math.pi], dtype=np.float32)
    return spaces.Box(-high, high, shape=(OBS_DIM[kind],)), spaces.Discrete(NUM_ACTIONS[kind])


class DeviceEnvBatch:
    def __init__(self, kind, num_envs: int, seed: int | None = None, env_base: int = 0, ring: int = STAT_RING):
        self.kind = KINDS[kind] if isinstance(kind, str) else int(kind)
        self.num_envs = int(num_envs)
        self.obs_dim = OBS_DIM[self.kind]
        self.seed = default_seed() if seed is None else int(seed)
        self.env_base = int(env_base)
        self.ring = int(ring)
        self.lib = _lib.load()
        dev = self.device = ops.device()
        N = self.num_envs
        self.state = torch.zeros((N, 4), dtype=torch.float64, device=dev)
        self.elapsed = torch.zeros(N, dtype=torch.int32, device=dev)
        self.ep_ret = torch.zeros(N, dtype=torch.float32, device=dev)
        self.ep_len = torch.zeros(N, dtype=torch.int32, device=dev)
        self.episode = torch.zeros(N, dtype=torch.int32, device=dev)  # u32 bit pattern
        self.ret_ring = torch.zeros((N, ring), dtype=torch.float32, device=dev)
        self.len_ring = torch.zeros((N, ring), dtype=torch.int32, device=dev)
        self.ring_cnt = torch.zeros(N, dtype=torch.int32, device=dev)
        self._ptrs = None

    def pointers(self):
        """Device addresses of (state, elapsed, ep_ret, ep_len, episode, ret_ring, len_ring, ring_cnt); the
        tensors are allocated once and only ever written in place."""
        if self._ptrs is None:
            self._ptrs = tuple(t.data_ptr() for t in (self.state, self.elapsed, self.ep_ret, self.ep_len, self.episode,
                                                      self.ret_ring, self.len_ring, self.ring_cnt))
        return self._ptrs

    def reset(self, obs_out: torch.Tensor | None = None, reset_draw: torch.Tensor | None = None) -> torch.Tensor:
        N = self.num_envs
        if obs_out is None:
            obs_out = torch.empty((N, self.obs_dim), dtype=torch.float32, device=self.device)
        rd = None if reset_draw is None else reset_draw.to(self.device, torch.float64).contiguous()
        rc = self.lib.ppo_env_reset(
            self.kind, self.state.data_ptr(), self.elapsed.data_ptr(), self.ep_ret.data_ptr(),
            self.ep_len.data_ptr(), self.episode.data_ptr(), ops._ptr(rd), self.seed, self.env_base,
            obs_out.data_ptr(), N, ops._stream())
        _lib.check(rc, "ppo_env_reset")
        return obs_out

    def step(self, action, obs_out, term_obs, reward, done, trunc, is_last: bool = False, reset_draw=None):
        """All arguments are device tensors written in place (rows of the rollout buffers)."""
        rd = None if reset_draw is None else reset_draw.to(self.device, torch.float64).contiguous()
        rc = self.lib.ppo_env_step(
            self.kind, self.state.data_ptr(), self.elapsed.data_ptr(), self.ep_ret.data_ptr(),
            self.ep_len.data_ptr(), self.episode.data_ptr(), action.data_ptr(), ops._ptr(rd), self.seed,
            self.env_base, obs_out.data_ptr(), ops._ptr(term_obs), reward.data_ptr(), done.data_ptr(),
            trunc.data_ptr(), self.ret_ring.data_ptr(), self.len_ring.data_ptr(), self.ring_cnt.data_ptr(),
            self.ring, self.num_envs, int(bool(is_last)), ops._stream())
        _lib.check(rc, "ppo_env_step")

    def episode_stats(self):
        """(returns, lengths) of the last <=ring finished episodes of every env, flattened:
        what the reference concatenates from env.return_queue / length_queue (src/ppo.py:407,415)."""
        cnt = self.ring_cnt.clamp(max=self.ring)
        mask = torch.arange(self.ring, device=self.device)[None, :] < cnt[:, None]
        return self.ret_ring[mask], self.len_ring[mask]


class DeviceEnv:
    """One env with the gym.Env protocol, backed by a 1-env DeviceEnvBatch (kernel launches
    with N = 1).  Used for probing spaces and for plumbing runs; SyncVecEnv recognises it
    and steps all of its envs as ONE batch instead of N of these."""

    def __init__(self, kind, seed: int | None = None, record_stats: bool = True):
        self.kind = KINDS[kind] if isinstance(kind, str) else int(kind)
        self.observation_space, self.action_space = make_spaces(self.kind)
        self.seed = seed
        self._batch = None
        if record_stats:  # RecordEpisodeStatistics attributes read at reference src/ppo.py:404-417
            self.return_queue = deque(maxlen=STAT_RING)
            self.length_queue = deque(maxlen=STAT_RING)

    def _b(self) -> DeviceEnvBatch:
        if self._batch is None:
            self._batch = DeviceEnvBatch(self.kind, 1, seed=self.seed)
            d, od = self._batch.device, self._batch.obs_dim
            self._obs = torch.empty((1, od), dtype=torch.float32, device=d)
            self._term = torch.zeros((1, od), dtype=torch.float32, device=d)
            self._rew = torch.empty(1, dtype=torch.float32, device=d)
            self._done = torch.empty(1, dtype=torch.int32, device=d)
            self._trunc = torch.empty(1, dtype=torch.int32, device=d)
            self._act = torch.empty(1, dtype=torch.int32, device=d)
        return self._batch

    def reset(self, **_kw):
        b = self._b()
        obs = b.reset(self._obs)
        return obs[0].cpu().numpy(), {}

    def step(self, action):
        b = self._b()
        self._act.fill_(int(np.asarray(action).reshape(-1)[0]))
        ret_before = float(b.ep_ret.item())
        len_before = int(b.ep_len.item())
        b.step(self._act, self._obs, self._term, self._rew, self._done, self._trunc, is_last=False)
        done, trunc = bool(self._done.item()), bool(self._trunc.item())
        reward = float(self._rew.item())
        ended = done or trunc
        obs = (self._term if ended else self._obs)[0].cpu().numpy()  # gym protocol: terminal obs, no auto-reset
        if ended and hasattr(self, "return_queue"):
            self.return_queue.append(np.float32(ret_before + reward))
            self.length_queue.append(len_before + 1)
        return obs, reward, done, trunc, {}

    def render(self):
        raise NotImplementedError("device-resident envs have no renderer (GIF recording is out of scope)")

    def close(self):
        self._batch = None
