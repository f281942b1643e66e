"""GPU: BASELINE configs[4] at its shape -- Ant-style continuous control, N = 4096 envs x T = 256, env stepping
delegated to the host (reference src/env_ant.py:6-9: MuJoCo Ant-v4, obs Box(27), actions Box(8) in [-1, 1]), the
Beta policy head (src/ppo.py:191-207, ScaledBeta :18-56) on PyTorch, GAE and the loss on the GPU kernels.
MuJoCo is not installed, so the env is a scripted stand-in with the gym.Env protocol.

  * ppo_gae_dense / ppo_gae_ragged at [256, 4096] against the oracle, bit for bit (synthetic inputs of SURVEY 8d);
  * one PPO.train update, 4 minibatches of T N / 4 = 262 144 samples, through ScaledBeta + ppo_loss_fwd_bwd +
    ppo_linear_wgrad: the first minibatch's losses against a float64 torch-autograd recomputation of src/ppo.py:
    595-604 to 1e-5, and its (clipped) gradients against that pass's.
"""
import copy
from collections import deque

import numpy as np
import pytest
import torch

from oracle import cref, pyref

pytestmark = pytest.mark.gpu

T, N = 256, 4096


def test_gae_at_the_config5_shape_is_bit_exact():
    from ppo_amd import ops
    g = torch.Generator().manual_seed(1234)
    r, v = torch.randn(T, N, generator=g), torch.randn(T, N, generator=g)
    done = (torch.rand(T, N, generator=g) < 1 / 50).to(torch.int32)
    trunc = ((torch.rand(T, N, generator=g) < 1 / 500) & (done == 0)).to(torch.int32)
    trunc[-1] = 1 - done[-1]
    E = int((done + trunc).sum())
    ev = torch.randn(E, generator=g)
    rn, vn, dn, tn, en = (x.numpy() for x in (r, v, done, trunc, ev))
    want = cref.gae_ragged(rn, vn, en, dn, tn, 0.99, 0.95)
    got = ops.gae_ragged(r.cuda(), v.cuda(), ev.cuda(), done.cuda(), trunc.cuda(), 0.99, 0.95)
    assert got.shape == (T, N) and np.array_equal(got.cpu().numpy(), want)
    dense = cref.ragged_to_dense(en, dn, tn)
    got_d, ret = ops.gae_dense(r.cuda(), v.cuda(), torch.from_numpy(dense).cuda(), done.cuda(), trunc.cuda(), 0.99, 0.95,
                               want_returns=True)
    assert np.array_equal(got_d.cpu().numpy(), want)
    assert np.array_equal(ret.cpu().numpy(), want + vn)
    # the allocation-free form agrees too (N <= its limit)
    got_n = ops.gae_ragged(r.cuda(), v.cuda(), ev.cuda(), done.cuda(), trunc.cuda(), 0.99, 0.95, noalloc=True)
    assert np.array_equal(got_n.cpu().numpy(), want)


class FakeAnt:
    """gym.Env protocol stand-in for Ant-v4 + RecordEpisodeStatistics: obs Box(27), actions Box(8) in [-1, 1]."""

    def __init__(self, seed):
        from ppo_amd import spaces
        self.observation_space = spaces.Box(-np.inf, np.inf, (27,))
        self.action_space = spaces.Box(-1.0, 1.0, (8,))
        self.rng = np.random.default_rng(seed)
        self.seed, self.k, self.ret = seed, 0, 0.0
        self.x = np.zeros(27, dtype=np.float32)
        self.return_queue, self.length_queue = deque(maxlen=100), deque(maxlen=100)

    def reset(self):
        self.k, self.ret = 0, 0.0
        self.x = (0.1 * self.rng.standard_normal(27)).astype(np.float32)
        return self.x, {}

    def step(self, action):
        a = np.asarray(action, dtype=np.float32)
        assert a.shape == (8,) and np.all(a >= -1.0) and np.all(a <= 1.0)
        self.k += 1
        self.x = (0.9 * self.x + 0.1 * np.tile(a, 4)[:27]).astype(np.float32)
        r = 1.0 - 0.5 * float(a @ a) + float(self.x[0])
        self.ret += r
        done = self.k == 40 + self.seed % 90                  # "unhealthy" terminations of 40..129 steps,
        trunc = (not done) and self.k == 100                  # a time limit at 100 for the others
        if done or trunc:
            self.return_queue.append(np.float32(self.ret)); self.length_queue.append(self.k)
        return self.x, r, done, trunc, {}


def test_box_head_trainer_update_at_the_config5_shape():
    from ppo_amd import PPO
    seeds = iter(range(100000))
    torch.manual_seed(0)
    ppo = PPO(0.99, 0.95, 1, 1, 0, 1, True)                   # dense joint network, norm_adv, clipped value loss
    B = T * N // 4
    rec = {}

    def hook(info):
        if "first" not in rec:
            rec["first"] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in info.items()}
            rec["grads"] = [p.grad.detach().clone() for p in ppo.agent.parameters()]
        rec["count"] = rec.get("count", 0) + 1

    ppo.minibatch_hook = hook
    orig = ppo.train_step

    def train_step(*a, **k):
        rec["init"] = copy.deepcopy(ppo.agent.state_dict())
        return orig(*a, **k)

    ppo.train_step = train_step
    ppo.train(lambda: FakeAnt(next(seeds)), num_updates=1, num_envs=N, steps_per_env=T, num_epochs=1, batch_size=B,
              critic_coef=0.5, entropy_coef=0.0, clip_ratio=0.2, max_grad_norm=0.5, learning_rate=3e-4)
    ve = ppo.vec_env
    assert not ve.native and rec["count"] == 4
    # reference tests/test_vec_env.py:74-81 contract at this shape
    assert ve.states.shape == (T, N, 27) and ve.states.dtype == torch.float32 and ve.states.requires_grad
    assert ve.actions.shape == (T, N, 8) and ve.actions.dtype == torch.float32
    assert ve.rewards.shape == ve.values.shape == ve.log_probs.shape == (T, N)
    assert ve.done_flags.dtype == torch.int32 and ve.trunc_flags.dtype == torch.int32
    ended = (ve.done_flags + ve.trunc_flags) > 0
    assert bool(ended[-1].all()) and ve.end_states.shape == (int(ended.sum()), 27)
    assert ppo.last_paths == {"fused_head": False, "fused_gather": True, "fused_update": False, "fused_optimizer": False}
    assert ve.median_ep_length == np.percentile(np.concatenate([np.array(e.length_queue) for e in ve.envs]), 50)

    # GAE on this rollout: the trainer's advantages are the oracle's, bit for bit
    with torch.no_grad():
        agent0 = copy.deepcopy(ppo.agent)
        agent0.load_state_dict(rec["init"])
        end_values = agent0.get_values(ve.end_states)
        adv = ppo.compute_advantages(ve.rewards, ve.values, end_values, ve.done_flags, ve.trunc_flags)
    want = cref.gae_ragged(ve.rewards.cpu().numpy(), ve.values.detach().cpu().numpy(), end_values.cpu().numpy(),
                           ve.done_flags.cpu().numpy(), ve.trunc_flags.cpu().numpy(), 0.99, 0.95)
    assert np.array_equal(adv.cpu().numpy(), want)

    # the first minibatch again, in float64 with torch autograd only (src/ppo.py:591-604 with compute_losses :490-527)
    first = rec["first"]
    idx = first["idx"].long()
    assert idx.numel() == B
    a64 = copy.deepcopy(agent0).double()
    flat = lambda x: x.detach().flatten(0, 1)
    st, ac = flat(ve.states)[idx].double(), flat(ve.actions)[idx].double()
    old_lp, old_v, mb_adv = (flat(x)[idx].double() for x in (ve.log_probs, ve.values, adv))
    _, new_lp, new_v, ent = a64.get_actions_and_values(st, ac)
    pl, vl, cf, kl = pyref.losses_torch(torch.exp(new_lp - old_lp), new_v, old_v, mb_adv, 0.2, norm_adv=1, clip_va_loss=1)
    for name, got, ref in (("policy_loss", first["policy_loss"], pl), ("critic_loss", first["critic_loss"], vl),
                           ("entropy", first["entropy"], ent), ("kl_div", first["kl_div"], kl)):
        assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref)) + 2e-7, (name, float(got), float(ref))
    ((pl + 0.5 * vl - 0.0 * ent) * 1.0).backward()
    g64 = [p.grad for p in a64.parameters()]
    norm = float(torch.sqrt(sum((g * g).sum() for g in g64)))
    coef = min(0.5 / (norm + 1e-6), 1.0)                      # clip_grad_norm_(max_grad_norm = 0.5), src/ppo.py:608
    for (name, _), got, ref in zip(ppo.agent.named_parameters(), rec["grads"], g64):
        ref = ref * coef
        err = float((got.double() - ref).abs().max())
        assert err <= 1e-4 * float(ref.abs().max()) + 1e-9, (name, err, float(ref.abs().max()))
