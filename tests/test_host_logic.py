"""CPU suite: host-side logic of the Python surface (no kernels): the host-stepped rollout
against the golden produced by the reference's SyncVecEnv.rollout (oracle/gen_golden.py),
Agent contracts of the reference's tests/test_agent.py, percentiles, CLI parsing."""
import os

import numpy as np
import pytest
import torch

from ppo_amd import Agent, ScaledBeta, SyncVecEnv, spaces
from ppo_amd.ppo import _percentiles


class ScriptedEnv:
    """Same deterministic fake env as oracle/gen_golden.py (obs = env id, episode, step)."""

    def __init__(self, eid, ep_len, trunc_at):
        self.eid, self.ep_len, self.trunc_at = eid, ep_len, trunc_at
        self.episode, self.k = -1, 0

    def _obs(self):
        return np.array([self.eid, self.episode, self.k, 0.5], np.float32)

    def reset(self):
        self.episode += 1
        self.k = 0
        return self._obs(), {}

    def step(self, action):
        self.k += 1
        done = (self.k == self.ep_len[self.episode % len(self.ep_len)])
        trunc = (self.k == self.trunc_at) and not done
        return self._obs(), float(self.eid + 1 + 0.25 * self.k), done, trunc, {}


def test_host_rollout_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_rollout.npz"))
    scheds = [(g["sched_len0"].tolist(), int(g["trunc_at"][0])), (g["sched_len1"].tolist(), int(g["trunc_at"][1])),
              (g["sched_len2"].tolist(), int(g["trunc_at"][2]))]
    it = iter(range(3))

    def env_fn():
        i = next(it)
        return ScriptedEnv(i, *scheds[i])

    agent = Agent(spaces.Box(-1.0, 1.0, (4,)), spaces.Discrete(2), conv_net=0, joint_net=1, device="cpu")
    ve = SyncVecEnv(env_fn, 3, 12, 0, 0.0, agent)
    assert not ve.native
    for k in range(2):
        ve.rollout()
        p = f"r{k}_"
        assert ve.states.requires_grad
        assert np.array_equal(ve.states.detach().numpy(), g[p + "states"])
        assert np.array_equal(ve.rewards.numpy(), g[p + "rewards"])
        assert np.array_equal(ve.done_flags.numpy(), g[p + "done"])
        assert np.array_equal(ve.trunc_flags.numpy(), g[p + "trunc"])
        # end_states: actor-major, time-ascending (reference src/ppo.py:401-402).  The golden was
        # produced with device="cpu", where the reference's `end_state.cpu()` (:379) is a no-op on
        # a VIEW of t_new_states that the reset two lines later overwrites (:382): its CPU path
        # therefore stores the RESET observation for every env that was reset, and the terminal
        # observation only for forced truncations.  On device="cuda" `.cpu()` copies and the
        # reference keeps the terminal observation -- the behaviour this build implements.
        ref_end, ours = g[p + "end_states"], ve.end_states.numpy()
        assert ours.shape == ref_end.shape
        done, trunc, states = g[p + "done"], g[p + "trunc"], g[p + "states"]
        row = 0
        for n in range(3):
            for t in range(12):
                if done[t, n] + trunc[t, n] == 0:
                    continue
                terminal = states[t, n] + np.array([0, 0, 1, 0], np.float32)   # scripted env: next step index
                assert np.array_equal(ours[row], terminal)
                forced = (t == 11 and done[t, n] == 0 and trunc[t, n] == 1 and
                          int(states[t, n, 2]) + 1 != scheds[n][1])
                if forced:
                    assert np.array_equal(ref_end[row], terminal)              # no reset: reference agrees
                else:
                    assert ref_end[row, 2] == 0 and ref_end[row, 1] == states[t, n, 1] + 1  # aliased reset obs
                row += 1
        assert row == ours.shape[0]
        assert np.array_equal(ve.t_states.numpy(), g[p + "t_states"])         # carry-over, no reset on forced trunc
        assert ve.total_return == np.float32(g[p + "total_return"])
        assert ve.global_steps == int(g[p + "global_steps"])
        assert ve.actions.dtype == torch.int32 and ve.done_flags.dtype == torch.int32


def all_in_range(t, low, high):
    return int(((t < low).int() + (t > high).int()).sum()) == 0


def test_init_dense():
    """reference tests/test_agent.py:17-66"""
    B = 1000
    agent = Agent(spaces.Box(-1.0, 1.0, (4,)), spaces.Discrete(4), conv_net=0, joint_net=1, device="cpu")
    a, lp, v, ent = agent.get_actions_and_values(torch.rand(B, 4) * 2 - 1, actions=None)
    assert a.size(0) == B and a.dtype == torch.int32 and all_in_range(a, 0, 3)
    assert lp.size(0) == B and lp.dtype == torch.float32 and v.size(0) == B and v.dtype == torch.float32
    assert ent.size() == torch.Size([]) and ent.dtype == torch.float32
    agent = Agent(spaces.Box(-1.0, 1.0, (6,)), spaces.Box(-10.0, 10.0, (4,)), conv_net=0, joint_net=1, device="cpu")
    a, lp, v, ent = agent.get_actions_and_values(torch.rand(B, 6) * 2 - 1, actions=None)
    assert a.shape == (B, 4) and a.dtype == torch.float32 and all_in_range(a, -10.0, 10.0)
    assert lp.shape == (B,) and v.shape == (B,) and ent.size() == torch.Size([])


def test_init_convolutional():
    """reference tests/test_agent.py:68-117 (smaller batch: CPU conv is slow)"""
    B = 64
    img = spaces.Box(0, 255, (84, 84, 3))
    agent = Agent(img, spaces.Discrete(3), conv_net=1, joint_net=1, device="cpu")
    a, lp, v, ent = agent.get_actions_and_values(torch.rand(B, 84, 84, 3) * 255, actions=None)
    assert a.shape == (B,) and a.dtype == torch.int32 and all_in_range(a, 0, 2)
    assert lp.dtype == torch.float32 and v.shape == (B,) and ent.size() == torch.Size([])
    agent = Agent(img, spaces.Box(-2.0, 4.0, (6,)), conv_net=1, joint_net=0, device="cpu")
    a, lp, v, ent = agent.get_actions_and_values(torch.rand(B, 84, 84, 3) * 255, actions=None)
    assert a.shape == (B, 6) and a.dtype == torch.float32 and all_in_range(a, -2.0, 4.0)


def test_agent_init_gains_and_split_nets():
    agent = Agent(spaces.Box(-1.0, 1.0, (4,)), spaces.Discrete(2), conv_net=0, joint_net=0, device="cpu")
    assert hasattr(agent, "va_backbone")
    w = agent.pi_backbone[0].weight  # orthogonal with gain sqrt(2): W^T W = 2 I (64x4)
    assert torch.allclose(w.t() @ w, 2 * torch.eye(4), atol=1e-4)
    assert torch.allclose(agent.policy.weight @ agent.policy.weight.t(), 1e-4 * torch.eye(2), atol=1e-7)
    assert torch.allclose(agent.critic.weight @ agent.critic.weight.t(), torch.eye(1), atol=1e-5)
    s = torch.randn(5, 4)
    assert torch.equal(agent.get_values(s), agent.get_actions_and_values(s, None)[2])


def test_scaled_beta_matches_closed_form():
    modes, prec = torch.tensor([[0.3, 0.8]]), torch.tensor([[4.0, 9.0]])
    lows, highs = torch.tensor([-2.0, 0.0]), torch.tensor([2.0, 5.0])
    d = ScaledBeta(modes, prec, lows, highs)
    x = torch.tensor([[0.5, 3.0]])
    beta = torch.distributions.Beta(modes * prec + 1, prec + 2 - (modes * prec + 1))
    want = (beta.log_prob((x - lows) / (highs - lows + 1e-6)) - torch.log(highs - lows)).sum(1)
    assert torch.allclose(d.log_prob(x), want)
    assert torch.allclose(d.entropy(), (beta.entropy() + torch.log(highs - lows)).sum(1))
    s = d.sample()
    assert ((s >= lows) & (s <= highs)).all()


def test_percentiles_match_numpy():
    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 1000):
        x = np.sort(rng.standard_normal(n))
        got = _percentiles(torch.from_numpy(x), (5.0, 50.0, 95.0)).numpy()
        assert np.allclose(got, np.percentile(x, [5.0, 50.0, 95.0]))


def test_cli_positional_order_matches_reference():
    from ppo_amd import ppo_cli
    argv = "exp env_cartpole 200 8 128 4 128 0.5 0.01 0.2 0.5 3e-4 0.99 0.95 1 1 0 0".split()
    a = ppo_cli.build_parser().parse_args(argv + ["--use_gpu", "1", "--target_div", "0.01"])
    assert (a.experiment_name, a.env_module, a.num_updates, a.num_envs, a.steps_per_env) == ("exp", "env_cartpole", 200, 8, 128)
    assert (a.num_epochs, a.batch_size, a.critic_coef, a.entropy_coef, a.clip_ratio) == (4, 128, 0.5, 0.01, 0.2)
    assert (a.max_grad_norm, a.learning_rate, a.discount_factor, a.gae_factor) == (0.5, 3e-4, 0.99, 0.95)
    assert (a.norm_adv, a.clip_va_loss, a.conv_net, a.joint_network, a.use_gpu, a.target_div) == (1, 1, 0, 0, 1, 0.01)
    assert a.render_every == 0 and a.render_fps == 0.0 and a.early_stop_reward is None
    assert callable(ppo_cli.resolve_env_fn("env_acrobot"))


def test_clip_adam_supported_is_host_logic():
    """Which optimisers the one-launch clip + Adam step takes over (the reference's optim.Adam(lr, eps=1e-5),
    src/ppo.py:732-733) -- decided on the host, no device needed."""
    import torch
    from ppo_amd import ops
    w = torch.nn.Parameter(torch.zeros(4))
    assert ops.clip_adam_supported(torch.optim.Adam([w], lr=3e-4, eps=1e-5))
    assert not ops.clip_adam_supported(torch.optim.Adam([w], weight_decay=0.1))
    assert not ops.clip_adam_supported(torch.optim.Adam([w], amsgrad=True))
    assert not ops.clip_adam_supported(torch.optim.Adam([w], maximize=True))
    assert not ops.clip_adam_supported(torch.optim.Adam([w], lr=torch.tensor(1e-3)))
    assert not ops.clip_adam_supported(torch.optim.Adam([{"params": [w]}, {"params": [torch.nn.Parameter(torch.zeros(2))]}]))
    assert not ops.clip_adam_supported(torch.optim.AdamW([w]))
    assert not ops.clip_adam_supported(torch.optim.SGD([w], lr=0.1))


# ---- job-wide episode statistics: the host-side narrowing of the distributed radix select -------------------
def _f2key(x32):
    """numpy restatement of csrc/stats.hip f2key (order-preserving f32 -> u32)."""
    u = np.asarray(x32, dtype=np.float32).view(np.uint32)
    return np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000)).astype(np.uint32)


def _select_like_the_ranks_do(returns, lengths):
    """The selection SyncVecEnv.job_wide_episode_stats runs, with the (all-reduced) histograms made here in numpy:
    level 0 = 12 key bits, levels 1 / 2 = 10 bits each under the prefixes still in play."""
    from ppo_amd.ppo import _bucket_of, _key_to_float, _lerp, _select_ranks
    kr, kl = _f2key(returns), _f2key(np.asarray(lengths, dtype=np.float32))
    n = kr.size
    ranks = _select_ranks(n)
    h0 = [np.bincount(kr >> 20, minlength=4096), np.bincount(kl >> 20, minlength=4096)]
    pref, resid = [], []
    for t, r in enumerate(ranks):
        b, rem = _bucket_of(np.cumsum(h0[0 if t < 7 else 1]), r)
        pref.append(b); resid.append(rem)
    for level in (1, 2):
        shift, pshift = (10, 20) if level == 1 else (0, 10)
        low = 0
        for t in range(9):
            k = kr if t < 7 else kl
            m = (k >> pshift) == pref[t]
            if level == 1:
                vals = (returns if t < 7 else np.asarray(lengths, dtype=np.float32))[m]
                low |= int((np.asarray(vals, dtype=np.float32).view(np.uint32) & 1023).any())
            hist = np.bincount((k[m] >> shift) & 1023, minlength=1024)
            b, resid[t] = _bucket_of(np.cumsum(hist), resid[t])
            pref[t] = (pref[t] << 10) | b
        if level == 1 and not low:
            pref = [(p << 10) | (0 if (p >> 21) & 1 else 1023) for p in pref]
            break
    v = _key_to_float(np.array(pref, dtype=np.uint64))
    return [v[6], _lerp(v[0], v[1], n, 0.05), _lerp(v[2], v[3], n, 0.50), _lerp(v[4], v[5], n, 0.95), _lerp(v[7], v[8], n, 0.50)]


@pytest.mark.parametrize("kind", ["integers", "negative_integers", "floats", "one", "two", "duplicates"])
def test_job_wide_selection_equals_numpy_percentile(kind):
    """reference src/ppo.py:407-417: np.max, np.percentile([5, 50, 95]) of all returns, np.percentile(50) of all lengths."""
    rng = np.random.default_rng(3)
    n = {"one": 1, "two": 2}.get(kind, 5000)
    if kind == "integers":
        rets = rng.integers(8, 500, n).astype(np.float32)
    elif kind == "negative_integers":
        rets = -rng.integers(60, 500, n).astype(np.float32)
    elif kind == "duplicates":
        rets = np.full(n, 500.0, dtype=np.float32); rets[:7] = 499.0
    else:
        rets = (rng.standard_normal(n) * 300).astype(np.float32)
    lens = rng.integers(1, 500, n).astype(np.int32)
    got = _select_like_the_ranks_do(rets, lens)
    want = [rets.max(), *np.percentile(rets.astype(np.float64), [5, 50, 95]), np.percentile(lens.astype(np.float64), 50)]
    assert np.array_equal(np.array(got, dtype=np.float64), np.array(want, dtype=np.float64)), (got, want)
    if kind in ("integers", "negative_integers", "duplicates"):
        # what SyncVecEnv.job_wide_episode_stats runs for the native envs: the (all-reduced) DIRECT histograms of
        # ppo_episode_stats_incremental_ints, one bin per integer
        from ppo_amd.ppo import _direct_select, _lib_const
        R0, RB, LB = (_lib_const("PPO_STATS_INTS_" + k) for k in ("RET_ZERO", "RET_BINS", "LEN_BINS"))
        h_ret = np.bincount(rets.astype(np.int64) + R0, minlength=RB)
        h_len = np.bincount(lens.astype(np.int64), minlength=LB)
        got = _direct_select(h_ret, h_len, R0)
        assert np.array_equal(np.array(got, dtype=np.float64), np.array(want, dtype=np.float64)), (got, want)
