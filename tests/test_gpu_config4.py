"""GPU: BASELINE configs[3] through the trainer -- host-stepped image envs (the shape of the
reference's Breakout pipeline, src/env_breakout.py:9-34: 84 x 84 frames, 4 stacked on the channel
axis, Discrete(4)), the Nature-CNN Agent (src/ppo.py:91-122) on PyTorch-ROCm, and the GPU GAE +
logits-loss kernels, 64 envs x T = 128, one update.  ALE is not installed, so the env is a scripted
stand-in with the gym.Env protocol (deterministic done / trunc schedule, random frames)."""
from collections import deque

import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


class FakeImageEnv:
    def __init__(self, seed):
        from ppo_amd import spaces
        self.observation_space = spaces.Box(0.0, 1.0, (84, 84, 4))
        self.action_space = spaces.Discrete(4)
        self.rng = np.random.default_rng(seed)
        self.seed, self.k, self.ret = seed, 0, 0.0
        self.return_queue, self.length_queue = deque(maxlen=100), deque(maxlen=100)

    def _frame(self):
        return self.rng.random((84, 84, 4), dtype=np.float32)

    def reset(self):
        self.k, self.ret = 0, 0.0
        return self._frame(), {}

    def step(self, action):
        assert 0 <= int(action) < 4
        self.k += 1
        r = float(np.sign(int(action) - 1.5))                 # reward clipping to {-1, +1} (src/env_breakout.py:33)
        self.ret += r
        done = self.k == 23 + self.seed % 7                   # scripted: episodes of 23..29 steps,
        trunc = (not done) and self.k == 27                   # some cut at 27 by a time limit
        if done or trunc:
            self.return_queue.append(np.float32(self.ret)); self.length_queue.append(self.k)
        return self._frame(), r, done, trunc, {}


def test_image_env_conv_agent_one_update():
    from ppo_amd import PPO
    seeds = iter(range(1000))
    torch.manual_seed(0)
    ppo = PPO(0.99, 0.95, 1, 1, 1, 1, True)                   # conv_net = 1, joint network
    N, T = 64, 128
    ppo.train(lambda: FakeImageEnv(next(seeds)), num_updates=1, num_envs=N, steps_per_env=T, num_epochs=1,
              batch_size=2048, critic_coef=0.5, entropy_coef=0.01, clip_ratio=0.1, max_grad_norm=0.5,
              learning_rate=2.5e-4, target_div=0.05)
    ve = ppo.vec_env
    assert not ve.native
    # reference tests/test_vec_env.py:33-37 / :55-62 style contract for image states
    assert ve.states.shape == (T, N, 84, 84, 4) and ve.states.dtype == torch.float32 and ve.states.requires_grad
    assert ve.actions.shape == (T, N) and ve.actions.dtype == torch.int32
    assert ve.done_flags.dtype == torch.int32 and ve.trunc_flags.dtype == torch.int32
    assert ve.t_states.shape == (N, 84, 84, 4)
    ended = (ve.done_flags + ve.trunc_flags) > 0
    assert bool(ended[-1].all()) and ve.end_states.shape == (int(ended.sum()), 84, 84, 4)
    assert int(ve.trunc_flags[:-1].sum()) > 0 and int(ve.done_flags.sum()) > 0
    # the GPU kernels were on the path: one-launch gather (wide rows) and the fused logits loss
    assert ppo.last_paths == {"fused_head": True, "fused_gather": True, "fused_update": False, "fused_optimizer": False}   # conv net: PyTorch MLP around the loss kernel
    m = ppo.last_metrics
    assert np.isfinite(m["losses/policy_loss"]) and np.isfinite(m["losses/critic_loss"]) and m["losses/entropy"] > 1.0
    assert ve.median_ep_length in (23.0, 24.0, 25.0, 26.0, 27.0)
    # compute_advantages on this rollout == the oracle's restatement of src/ppo.py:452-488, bit for bit
    with torch.no_grad():
        end_values = ppo.agent.get_values(ve.end_states)
        adv = ppo.compute_advantages(ve.rewards, ve.values, end_values, ve.done_flags, ve.trunc_flags)
    want = cref.gae_ragged(ve.rewards.cpu().numpy(), ve.values.detach().cpu().numpy(), end_values.cpu().numpy(),
                           ve.done_flags.cpu().numpy(), ve.trunc_flags.cpu().numpy(), 0.99, 0.95)
    assert adv.shape == (T, N) and adv.dtype == torch.float32
    assert np.array_equal(adv.cpu().numpy(), want)
