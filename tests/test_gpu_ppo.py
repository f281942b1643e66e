"""GPU: the Python surface (PPO / SyncVecEnv / Agent) on the kernels -- the reference's
tests/test_vec_env.py and tests/test_ppo.py::test_cartpole contracts, plus rollout parity of
the device path against the oracle's serial restatement of the reference loop."""
import time

import numpy as np
import pytest
import torch

from oracle import cref, pyref
from ppo_amd import _lib

pytestmark = pytest.mark.gpu


def test_vec_env_states():
    """reference tests/test_vec_env.py:11-24 (CartPole part; ALE is not installed)"""
    from ppo_amd import Agent, SyncVecEnv
    from ppo_amd.env_cartpole import train_fn as env_fn
    test_env = env_fn()
    agent = Agent(test_env.observation_space, test_env.action_space, conv_net=0, joint_net=1, device="cpu")
    vec_env = SyncVecEnv(env_fn, num_envs=6, steps_per_env=125, render_every=0, render_fps=0.0, agent=agent)
    states = vec_env.vec_reset()
    assert states.size() == torch.Size([6, 4])
    assert states.dtype == torch.float32


@pytest.mark.parametrize("device", ["cpu", "cuda"])
def test_vec_env_rollout(device):
    """reference tests/test_vec_env.py:39-62"""
    from ppo_amd import Agent, SyncVecEnv
    from ppo_amd.env_cartpole import train_fn as env_fn

    def matching_shape_dtype(x, shape, dtype):
        assert x.size() == torch.Size(shape)
        assert x.dtype == dtype

    test_env = env_fn()
    agent = Agent(test_env.observation_space, test_env.action_space, conv_net=0, joint_net=1, device=device)
    vec_env = SyncVecEnv(env_fn, num_envs=2, steps_per_env=125, render_every=0, render_fps=0.0, agent=agent)
    vec_env.rollout()
    assert vec_env.states.requires_grad
    matching_shape_dtype(vec_env.states, (125, 2, 4), torch.float32)
    matching_shape_dtype(vec_env.actions, (125, 2), torch.int32)
    matching_shape_dtype(vec_env.rewards, (125, 2), torch.float32)
    matching_shape_dtype(vec_env.done_flags, (125, 2), torch.int32)
    matching_shape_dtype(vec_env.trunc_flags, (125, 2), torch.int32)
    matching_shape_dtype(vec_env.values, (125, 2), torch.float32)
    matching_shape_dtype(vec_env.log_probs, (125, 2), torch.float32)
    # rollout invariants of reference src/ppo.py:352-356, :377-379, :401-402
    ended = (vec_env.done_flags + vec_env.trunc_flags) > 0
    assert bool(ended[-1].all())
    assert vec_env.end_states.shape == (int(ended.sum()), 4)
    assert vec_env.global_steps == 250 and vec_env.total_return == np.float32(250.0)


@pytest.mark.parametrize("whole", [None, "f32", "f16_split", "f16_split_wave64"],
                         ids=["per_step_kernels", "whole_rollout_f32", "whole_rollout_f16_split_wave32",
                              "whole_rollout_f16_split_wave64"])
@pytest.mark.parametrize("joint", [1, 0], ids=["joint", "split"])
@pytest.mark.parametrize("kind", ["cartpole", "acrobot"])
def test_device_rollout_matches_serial_reference_loop(kind, joint, whole, request):
    """Same policy, same sampled actions (replayed), same reset draws (Philox): the fused device
    rollout and the oracle's restatement of the reference's serial loop (src/ppo.py:320-402) give
    the same buffers -- flags exact, the rest to f32 tolerance -- over two consecutive rollouts."""
    from ppo_amd import Agent, SyncVecEnv
    from ppo_amd.envs import DeviceEnv
    torch.manual_seed(0)
    f16 = whole in ("f16_split", "f16_split_wave64")
    N, T, seed = (70 if f16 else 64), 96, 2024   # 70: a partial wave (tail lanes shadow env N-1)
    probe = DeviceEnv(kind)
    agent = Agent(probe.observation_space, probe.action_space, conv_net=0, joint_net=joint, device="cuda")
    with torch.no_grad():  # away from the near-uniform init so that log-probs / values are non-trivial
        agent.policy.weight.mul_(60.0)
        agent.policy.bias.uniform_(-0.5, 0.5)
        agent.critic.bias.fill_(0.3)
    ve = SyncVecEnv(lambda: DeviceEnv(kind, seed=seed), N, T, 0, 0.0, agent, seed=seed)
    from ppo_amd import ops
    ve.fused_rollout = whole is not None
    ve.net_mode = ops.NET_F32 if whole == "f32" else ops.NET_F16_SPLIT
    # small batches run 32 envs per wave (lanes 0-31); force the 64-env form for the last case
    ve.wave_envs = 64 if whole == "f16_split_wave64" else 0

    ora = cref.BatchEnv(kind, N, seed=seed)
    t_states = torch.from_numpy(ora.reset())
    assert np.allclose(ve.t_states.cpu().numpy(), t_states.numpy(), atol=1e-7)

    class OracleEnv:  # gym.Env protocol view of one oracle env, resets drawn from the same Philox stream
        def __init__(self, i):
            self.i = i
            self.b = cref.BatchEnv(kind, 1, seed=seed, env_base=i)
            self.b.reset()

        def reset(self):
            return self.b.reset()[0], {}

        def step(self, a):
            # step WITHOUT the fused bookkeeping: plain env.step semantics
            st, el = self.b.state.copy(), self.b.elapsed.copy()
            import ctypes as C
            obs = np.empty(self.b.od, np.float32); r = C.c_float()
            fn = cref.lib().oracle_cartpole_step if kind == "cartpole" else cref.lib().oracle_acrobot_step
            f = fn(self.b.state.ctypes.data_as(C.POINTER(C.c_double)), self.b.elapsed.ctypes.data_as(C.POINTER(C.c_int)),
                   C.c_int(int(a)), obs.ctypes.data_as(C.POINTER(C.c_float)), C.byref(r))
            return obs, r.value, bool(f & 1), bool(f & 2), {}

    envs = [OracleEnv(i) for i in range(N)]
    for k in range(2):
        ve.rollout()
        acts = ve.actions.cpu()
        step = {"t": 0}

        def policy(states):
            t = step["t"]; step["t"] += 1
            a = acts[t]
            lp, v = ve.log_probs[t].cpu(), ve.values[t].cpu()
            return a, lp, v

        ref = pyref.serial_rollout(envs, t_states, policy, T)
        t_states = ref["t_states"]
        assert torch.equal(ve.done_flags.cpu(), ref["done"]) and torch.equal(ve.trunc_flags.cpu(), ref["trunc"])
        assert torch.equal(ve.rewards.cpu(), ref["rewards"])
        assert float(ve.total_return) == float(ref["rewards"].sum(dtype=torch.float64))   # integer-valued: exact
        assert torch.allclose(ve.states.detach().cpu(), ref["states"], rtol=1e-5, atol=1e-6)
        assert ve.end_states.shape == ref["end_states"].shape
        assert torch.allclose(ve.end_states.cpu(), ref["end_states"], rtol=1e-5, atol=1e-6)
        assert torch.allclose(ve.t_states.cpu(), t_states, rtol=1e-5, atol=1e-6)
        # the policy outputs recorded by the fused sampler are consistent with the agent
        # policy outputs recorded by the rollout vs the PyTorch forward of the same agent on the
        # recorded states (all T*N samples): 1e-5 relative, abs floor 2e-6 (f32 summation order)
        with torch.no_grad():
            logits, vals = agent.policy_and_values(ve.states.detach().flatten(0, 1))
        lp = torch.log_softmax(logits, 1).gather(1, ve.actions.flatten().long()[:, None])[:, 0]
        assert torch.allclose(lp, ve.log_probs.flatten(), rtol=1e-5, atol=2e-6)
        assert torch.allclose(vals, ve.values.flatten(), rtol=1e-5, atol=2e-6)
        # and the sampled actions are the Philox draw applied to those logits: against the per-step sampler
        # kernel, and exactly -- outside a 1e-5 band around the CDF boundaries, where an ulp in a logit
        # may flip the draw -- against the oracle's draw applied to the PyTorch logits in float64
        from ppo_amd import ops
        for t in (0, T // 2, T - 1):
            with torch.no_grad():
                lg, _ = agent.policy_and_values(ve.states.detach()[t])
            a_ref, _ = ops.categorical_sample(lg, seed, 0, k * T + t)
            assert (a_ref == ve.actions[t]).float().mean() > 0.98
            u = cref.sample_uniform(seed, 0, k * T + t, N)
            bad, amb = cref.categorical_check(lg.cpu().numpy(), u, ve.actions[t].cpu().numpy())
            assert bad == 0 and amb <= 2, (t, bad, amb)
        assert int(ref["done"].sum()) > 0 or kind == "acrobot"


def test_train_step_advantages_dense_equals_reference_form():
    """PPO.train_step's dense-bootstrap GAE equals the drop-in ragged compute_advantages."""
    from ppo_amd import PPO
    from ppo_amd.env_cartpole import train_fn
    ppo = PPO(0.99, 0.95, 1, 1, 0, 0, True, seed=3)
    ppo.train(train_fn, num_updates=1, num_envs=256, steps_per_env=64, num_epochs=1, batch_size=4096,
              critic_coef=0.5, entropy_coef=0.01, clip_ratio=0.2, max_grad_norm=0.5, learning_rate=3e-4)
    ve = ppo.vec_env
    with torch.no_grad():
        ev = ppo.agent.get_values(ve.end_states)
        dense = ppo._advantages(ve, ev)
        ragged = ppo.compute_advantages(ve.rewards, ve.values, ev, ve.done_flags, ve.trunc_flags)
    assert torch.equal(dense, ragged)
    m = ppo.last_metrics
    assert set(m) >= {"utils/env_steps_per_sec", "losses/policy_loss", "metrics/kl_div", "metrics/ep_return_max"}
    assert np.isfinite(m["losses/policy_loss"]) and m["utils/env_steps_per_sec"] > 0


def test_cartpole():
    """reference tests/test_ppo.py:124-159: same hyper-parameters, same thresholds."""
    from ppo_amd import PPO
    from ppo_amd.env_cartpole import train_fn as env_fn
    ppo = PPO(discount_factor=0.99, gae_factor=0.95, norm_adv=1, clip_va_loss=1, conv_net=0, joint_network=0,
              use_gpu=True)
    time_start = time.time()
    ppo.train(env_fn=env_fn, num_updates=200, num_envs=4, steps_per_env=125, num_epochs=4, batch_size=128,
              critic_coef=0.5, entropy_coef=0.01, clip_ratio=0.2, max_grad_norm=0.5, learning_rate=3e-4,
              target_div=0.01)
    train_time_limit = 60
    ep_return_threshold = 250
    assert time.time() - time_start < train_time_limit
    assert ppo.vec_env.max_ep_return > ep_return_threshold


def test_cli_plumbing_config1():
    """BASELINE config 1: CartPole, 8 envs x 128 steps through the CLI's positional arguments."""
    from ppo_amd import ppo_cli
    argv = "plumb env_cartpole 2 8 128 4 128 0.5 0.01 0.2 0.5 3e-4 0.99 0.95 1 1 0 0 --use_gpu 1".split()
    agent = ppo_cli.main(argv)
    assert agent.updates == 2 and agent.vec_env.num_envs == 8 and agent.last_metrics["utils/env_steps_per_sec"] > 0


@pytest.mark.parametrize("joint", [1, 0])
def test_checkpoint_resume_is_bit_identical(tmp_path, monkeypatch, joint):
    """Train 6 updates straight vs 3 updates -> checkpoint -> fresh process state -> resume 3 more:
    identical parameters, env state and metrics (the reference has no checkpointing)."""
    from ppo_amd import ppo_cli
    monkeypatch.setenv("PPO_AMD_SEED", "17")      # env seed of the first run (unseeded runs draw os.urandom)
    base = f"ck env_cartpole 6 64 32 2 512 0.5 0.01 0.2 0.5 3e-4 0.99 0.95 1 1 0 {joint} --use_gpu 1".split()

    def run(extra, seed=5):
        torch.manual_seed(seed)
        torch.cuda.manual_seed(seed)
        return ppo_cli.main(base + extra)

    straight = run([])
    ck3 = str(tmp_path / "ck3.pt")

    class Stop(Exception):
        pass

    # stop a run right after its update-3 checkpoint, as a killed job would
    from ppo_amd import PPO
    orig = PPO.save_checkpoint

    def save_and_stop(self, path):
        orig(self, path)
        raise Stop

    PPO.save_checkpoint = save_and_stop
    try:
        with pytest.raises(Stop):
            run(["--checkpoint_path", ck3, "--checkpoint_every", "3"])
    finally:
        PPO.save_checkpoint = orig
    torch.manual_seed(999)             # a resumed run depends on neither the ambient RNG state ...
    torch.cuda.manual_seed(999)
    monkeypatch.setenv("PPO_AMD_SEED", "4242")    # ... nor the env seed of the new process
    resumed = ppo_cli.main(base + ["--resume_from", ck3])
    assert resumed.updates == 6
    for a, b in zip(straight.agent.parameters(), resumed.agent.parameters()):
        assert torch.equal(a, b)
    for name in ("state", "elapsed", "episode", "ep_ret", "ring_cnt", "ret_ring"):
        assert torch.equal(getattr(straight.vec_env.batch, name), getattr(resumed.vec_env.batch, name)), name
    for k, v in straight.last_metrics.items():
        if not k.endswith("_per_sec"):
            assert resumed.last_metrics[k] == v or (np.isnan(v) and np.isnan(resumed.last_metrics[k])), k


def test_episode_statistics_match_numpy_percentiles():
    """ppo_episode_stats (radix select over the rings) vs the reference's formulas (src/ppo.py:407-417):
    np.max / np.percentile([5, 50, 95]) over the concatenated return deques, np.percentile(50) of lengths."""
    from ppo_amd import Agent, SyncVecEnv
    from ppo_amd.envs import DeviceEnv
    rng = np.random.default_rng(0)
    N = 1000
    probe = DeviceEnv("cartpole")
    agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
    ve = SyncVecEnv(lambda: DeviceEnv("cartpole"), N, 8, 0, 0.0, agent, seed=1)
    assert np.isnan(ve.max_ep_return) and np.isnan(ve.median_ep_length)
    ve._device_episode_stats()                       # no finished episode yet: stays NaN
    assert np.isnan(ve.max_ep_return) and np.isnan(ve.lower_ep_return)
    b = ve.batch
    cnt = rng.integers(0, 250, N).astype(np.int32)   # some rings empty, some wrapped (> 100)
    cnt[:10] = 0
    ret = rng.normal(100, 40, (N, 100)).astype(np.float32)
    ln = rng.integers(1, 500, (N, 100)).astype(np.int32)
    b.ring_cnt.copy_(torch.from_numpy(cnt)); b.ret_ring.copy_(torch.from_numpy(ret)); b.len_ring.copy_(torch.from_numpy(ln))
    ve._device_episode_stats()
    valid = np.arange(100)[None, :] < np.minimum(cnt, 100)[:, None]
    want = np.percentile(ret[valid].astype(np.float64), [5.0, 50.0, 95.0])
    assert np.allclose([ve.lower_ep_return, ve.median_ep_return, ve.upper_ep_return], want, rtol=1e-12)
    assert ve.max_ep_return == np.float32(ret[valid].max())
    assert ve.median_ep_length == np.percentile(ln[valid], 50.0)
    # running max survives a pass with smaller returns; percentiles follow the rings
    b.ret_ring.mul_(0.5)
    ve._device_episode_stats()
    assert ve.max_ep_return == np.float32(ret[valid].max())
    assert np.allclose(ve.median_ep_return, np.percentile((ret * 0.5)[valid].astype(np.float64), 50.0), rtol=1e-12)


@pytest.mark.parametrize("case", ["one", "two", "ints", "negative", "constant", "wide", "big"])
def test_episode_statistics_select_edge_cases(case):
    """The radix select on its hard inputs: 1-2 entries, heavy duplicates (CartPole's integer returns),
    negative values (Acrobot), all-equal, a wide dynamic range with mixed signs, and C2 scale."""
    from ppo_amd import Agent, SyncVecEnv
    from ppo_amd.envs import DeviceEnv
    rng = np.random.default_rng(5)
    N = 65536 if case == "big" else 777
    probe = DeviceEnv("cartpole")
    agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
    ve = SyncVecEnv(lambda: DeviceEnv("cartpole"), N, 4, 0, 0.0, agent, seed=1)
    cnt = rng.integers(0, 300, N).astype(np.int32)
    ret = rng.normal(0, 1, (N, 100)).astype(np.float32)
    if case in ("one", "two"):
        cnt[:] = 0
        cnt[5] = 1
        if case == "two":
            cnt[700] = 1
    elif case == "ints":
        ret = rng.integers(8, 60, (N, 100)).astype(np.float32)
    elif case == "negative":
        ret = -rng.integers(60, 500, (N, 100)).astype(np.float32)
    elif case == "constant":
        ret[:] = 9.0
    elif case == "wide":
        ret = (rng.standard_normal((N, 100)) * 10.0 ** rng.integers(-20, 20, (N, 100))).astype(np.float32)
        ret[3, 0] = 0.0
        ret[4, 0] = -0.0
    ln = rng.integers(1, 501, (N, 100)).astype(np.int32)
    b = ve.batch
    b.ring_cnt.copy_(torch.from_numpy(cnt)); b.ret_ring.copy_(torch.from_numpy(ret)); b.len_ring.copy_(torch.from_numpy(ln))
    ve._device_episode_stats()
    valid = np.arange(100)[None, :] < np.minimum(cnt, 100)[:, None]
    want = np.percentile(ret[valid].astype(np.float64), [5.0, 50.0, 95.0])
    got = np.array([ve.lower_ep_return, ve.median_ep_return, ve.upper_ep_return])
    assert np.allclose(got, want, rtol=1e-12, atol=0.0), (got, want)
    assert ve.max_ep_return == np.float32(ret[valid].max())
    assert ve.median_ep_length == np.percentile(ln[valid], 50.0)


@pytest.mark.parametrize("wave_envs", [32, 64])
@pytest.mark.parametrize("joint", [1, 0], ids=["joint", "split"])
@pytest.mark.parametrize("kind", ["cartpole", "acrobot"])
def test_fused_rollout_bootstraps_time_limit_truncations(kind, joint, wave_envs, request):
    """TimeLimit truncations in the middle of a rollout (reference: gymnasium TimeLimit + src/ppo.py:377-379,
    :555, :478): the whole-rollout kernel raises trunc at elapsed == 500, records the terminal observation,
    resets the env, and writes V(terminal observation) from its own second network pass.  Checked against
    the PyTorch critic on the recorded terminal observations."""
    from ppo_amd import Agent, SyncVecEnv
    from ppo_amd.envs import DeviceEnv
    torch.manual_seed(5)
    N, T = 200, 64
    probe = DeviceEnv(kind)
    agent = Agent(probe.observation_space, probe.action_space, 0, joint, "cuda")
    with torch.no_grad():
        agent.critic.bias.fill_(0.7)
    ve = SyncVecEnv(lambda: DeviceEnv(kind, seed=9), N, T, 0, 0.0, agent, seed=9)
    ve.wave_envs = wave_envs
    rng = np.random.default_rng(0)
    left = rng.integers(1, T - 1, N).astype(np.int32)          # steps until the time limit, per env
    ve.batch.elapsed.copy_(torch.from_numpy(500 - left))
    ve.rollout()
    tr, dn = ve.trunc_flags.cpu().numpy(), ve.done_flags.cpu().numpy()
    alive = np.array([dn[:left[n] - 1, n].sum() == 0 for n in range(N)])   # not terminated before its limit
    assert alive.sum() > (0 if kind == "cartpole" else N // 2)
    for n in np.nonzero(alive)[0]:
        assert tr[left[n] - 1, n] == 1                            # the step that reaches elapsed == 500
    mid = torch.from_numpy(tr[:-1] > 0).cuda()
    assert int(mid.sum()) >= int(alive.sum())
    term = ve._term_obs[:-1][mid]
    with torch.no_grad():
        want = agent.get_values(term)
    got = ve._ev_dense[:-1][mid]
    assert torch.allclose(got, want, rtol=1e-5, atol=2e-6)
    # the forced truncation of the last row is bootstrapped as well
    last = ve.trunc_flags[-1] > 0
    with torch.no_grad():
        want_last = agent.get_values(ve._term_obs[-1][last])
    assert torch.allclose(ve._ev_dense[-1][last], want_last, rtol=1e-5, atol=2e-6)
    # and the env restarted: the row after a truncation holds a fresh reset observation
    nxt = ve.states.detach()[1:][mid]
    bound = 0.05 if kind == "cartpole" else 1.0
    assert float(nxt.abs().max()) <= bound + 1e-6
    if kind == "acrobot":
        assert float(nxt[:, 4:].abs().max()) <= 0.1 + 1e-6          # velocities drawn from U(-0.1, 0.1)


def test_fused_rollout_settles_repeated_truncations_of_one_env():
    """A rollout longer than two time limits: an env hits the 500-step limit, then hits it again (or
    reaches the forced truncation of the last row) while the first bootstrap value is still pending --
    the kernel has to settle the earlier one first.  Every truncated cell gets V(terminal observation)."""
    from ppo_amd import Agent, SyncVecEnv
    from ppo_amd.envs import DeviceEnv
    torch.manual_seed(6)
    N, T = 96, 1100
    probe = DeviceEnv("acrobot")
    agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
    with torch.no_grad():
        agent.critic.bias.fill_(-0.4)
    ve = SyncVecEnv(lambda: DeviceEnv("acrobot", seed=4), N, T, 0, 0.0, agent, seed=4)
    ve.rollout()
    tr = ve.trunc_flags > 0
    per_env = tr.sum(0)
    assert int(per_env.max()) >= 3                    # two time limits and the last row, for some env
    with torch.no_grad():
        want = agent.get_values(ve._term_obs[tr])
    assert torch.allclose(ve._ev_dense[tr], want, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("kind,joint", [("cartpole", 1), ("acrobot", 0), ("acrobot", 1)])
def test_rollout_is_bit_identical_for_both_wave_widths(kind, joint, request):
    """32 envs per wave (small batches) and 64 envs per wave run the same arithmetic per env: every
    rollout buffer is bit-identical.  (Acrobot's 32-env form also splits each pair of sincos calls between the
    two lanes that carry an env, csrc/env_core.h Acrobot::trig_of<PAIR>; 333 envs leave a partly filled last wave.)"""
    from ppo_amd import Agent, SyncVecEnv
    from ppo_amd.envs import DeviceEnv
    torch.manual_seed(3)
    probe = DeviceEnv(kind)
    agent = Agent(probe.observation_space, probe.action_space, 0, joint, "cuda")
    with torch.no_grad():
        agent.policy.weight.mul_(40.0)
    got = {}
    for width in (32, 64):
        ve = SyncVecEnv(lambda: DeviceEnv(kind, seed=11), 333, 64, 0, 0.0, agent, seed=11)
        ve.wave_envs = width
        ve.rollout(); ve.rollout()
        got[width] = [getattr(ve, k).detach().clone() for k in
                      ("states", "actions", "log_probs", "values", "rewards", "done_flags", "trunc_flags", "t_states")]
        got[width].append(ve.end_states.clone())
    for a, b in zip(got[32], got[64]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("kind,joint", [("cartpole", 0), ("cartpole", 1), ("acrobot", 0), ("acrobot", 1)])
def test_rollout_full_form_matches_the_general_form(kind, joint, request):
    """A batch whose 64-env waves are all full (N a multiple of the 256-env workgroup) runs k_rollout's FULL form, which
    has no per-lane `live` branch (csrc/rollout.hip launch_net); any other N runs the general form.  An env's stream
    depends on its index and the seed only, so the first 500 envs of a 512-env batch and a 500-env batch must agree
    bit for bit in every rollout buffer, over two rollouts (the second starts from carried state)."""
    from ppo_amd import Agent, SyncVecEnv
    from ppo_amd.envs import DeviceEnv
    torch.manual_seed(5)
    probe = DeviceEnv(kind)
    agent = Agent(probe.observation_space, probe.action_space, 0, joint, "cuda")
    with torch.no_grad():
        agent.policy.weight.mul_(40.0)
    got = {}
    for N in (512, 500):
        ve = SyncVecEnv(lambda: DeviceEnv(kind, seed=13), N, 64, 0, 0.0, agent, seed=13)
        ve.wave_envs = 64
        ve.rollout(); ve.rollout()
        got[N] = [getattr(ve, k).detach()[:, :500].clone() for k in
                  ("states", "actions", "log_probs", "values", "rewards", "done_flags", "trunc_flags")]
        got[N].append(ve.t_states[:500].clone())
        es = ve.end_states                           # actor-major list of terminal observations
        got[N].append(es[ve._end_n < 500].clone())
    assert int(got[512][5].sum()) > 0 or kind == "acrobot"   # CartPole episodes ended inside the window
    for a, b in zip(got[512], got[500]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("ring,N", [(1, 300), (7, 1000), (63, 129), (64, 64), (65, 500), (100, 70), (130, 257), (257, 33)])
def test_episode_statistics_any_ring_length(ring, N):
    """ppo_episode_stats through the C ABI for ring lengths other than the 100 SyncVecEnv uses (the
    flat walk steps its (ring, index) pairs by 64 = q ring + r): shorter than a wave, equal to it,
    longer, not a multiple of anything."""
    from ppo_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(ring * 1000 + N)
    cnt = rng.integers(0, 3 * ring + 2, N).astype(np.int32)
    cnt[rng.integers(0, N, 5)] = 0
    ret = rng.integers(-40, 60, (N, ring)).astype(np.float32) * 0.5
    ln = rng.integers(1, 501, (N, ring)).astype(np.int32)
    d_ret, d_len, d_cnt = (torch.from_numpy(x).cuda() for x in (ret, ln, cnt))
    stats5 = torch.full((5,), float("nan"), dtype=torch.float64, device="cuda")
    ws = torch.zeros(lib.ppo_episode_stats_workspace_bytes(), dtype=torch.uint8, device="cuda")   # zeroed once
    for _ in range(2):                               # the second call runs on the workspace as the first left it
        rc = lib.ppo_episode_stats(d_ret.data_ptr(), d_len.data_ptr(), d_cnt.data_ptr(), ring, N, stats5.data_ptr(),
                                   ws.data_ptr(), ws.numel(), None)
        assert rc == 0
    torch.cuda.synchronize()
    assert int(ws[512:].count_nonzero()) == 0        # histograms cleared for the next call
    got = stats5.cpu().numpy()
    valid = np.arange(ring)[None, :] < np.minimum(cnt, ring)[:, None]
    want = np.percentile(ret[valid].astype(np.float64), [5.0, 50.0, 95.0])
    assert got[0] == ret[valid].max()
    assert np.allclose(got[1:4], want, rtol=1e-12, atol=0.0), (got, want)
    assert got[4] == np.percentile(ln[valid], 50.0)


class _FakeBoxEnv:
    """Host-stepped gym.Env-protocol env with a Box action space (the shape of BASELINE configs 4-5:
    env stepping stays on the host, only GAE and the loss run on the GPU kernels)."""

    def __init__(self, seed):
        from ppo_amd import spaces
        from collections import deque
        self.observation_space = spaces.Box(-np.inf, np.inf, (27,))
        self.action_space = spaces.Box(-1.0, 1.0, (8,))
        self.rng = np.random.default_rng(seed)
        self.k = 0
        self.return_queue, self.length_queue = deque(maxlen=100), deque(maxlen=100)
        self._ret = 0.0

    def reset(self):
        self.k, self._ret = 0, 0.0
        return self.rng.standard_normal(27).astype(np.float32), {}

    def step(self, action):
        assert np.asarray(action).shape == (8,)
        self.k += 1
        r = float(-np.square(action).sum())
        self._ret += r
        done, trunc = self.k == 17, self.k == 40
        if done or trunc:
            self.return_queue.append(np.float32(self._ret)); self.length_queue.append(self.k)
        return self.rng.standard_normal(27).astype(np.float32), r, done, trunc, {}


@pytest.mark.parametrize("use_gpu", [True, False])
def test_host_stepped_env_uses_gpu_gae_and_loss(use_gpu):
    """reference tests/test_vec_env.py:64-81 shapes (Ant: states (T,3,27), actions (T,3,8) f32) with a
    stand-in env, and one full PPO update through ppo_gae_ragged / ppo_loss_fwd_bwd."""
    from ppo_amd import PPO
    seeds = iter(range(100))
    ppo = PPO(0.99, 0.95, 1, 1, 0, 0, use_gpu)
    ppo.train(lambda: _FakeBoxEnv(next(seeds)), num_updates=2, num_envs=3, steps_per_env=50, num_epochs=2,
              batch_size=64, critic_coef=0.5, entropy_coef=0.01, clip_ratio=0.2, max_grad_norm=0.5,
              learning_rate=3e-4, target_div=0.05)
    ve = ppo.vec_env
    assert not ve.native and ve.states.requires_grad
    assert ve.states.shape == (50, 3, 27) and ve.actions.shape == (50, 3, 8) and ve.actions.dtype == torch.float32
    assert ve.rewards.shape == (50, 3) and ve.done_flags.dtype == torch.int32 and ve.trunc_flags.dtype == torch.int32
    assert ve.states.device.type == ("cuda" if use_gpu else "cpu")
    ended = (ve.done_flags + ve.trunc_flags) > 0
    assert ve.end_states.shape == (int(ended.sum()), 27) and bool(ended[-1].all())
    assert np.isfinite(ppo.last_metrics["losses/critic_loss"]) and ve.median_ep_length == 17


def test_native_both_flags_cell_through_dense_gae():
    """CartPole falling on the very step that reaches the 500-step limit: gymnasium reports
    terminated AND truncated, and both flags reach the reference's GAE (src/ppo.py:476-480), where
    `(1 - terminations)` becomes -1 and the ragged index is decremented by 2.  The whole-rollout
    kernel emits both flags natively, records the terminal observation once, writes its bootstrap
    value (trunc != 0), and the trainer's dense GAE follows the reference's (1 - 2) arithmetic on that
    cell: bit-exact against the oracle, and bit-equal to the reference's ragged form on the same
    rollout.  (The extra index decrement can only reach an EARLIER truncated cell of the same env,
    i.e. a second time-limit hit: impossible inside a rollout of T <= 500 steps.)"""
    from ppo_amd import Agent, SyncVecEnv, ops
    from ppo_amd.envs import DeviceEnv
    torch.manual_seed(2)
    N, T = 256, 16
    probe = DeviceEnv("cartpole")
    agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
    with torch.no_grad():
        agent.critic.bias.fill_(0.5)
    ve = SyncVecEnv(lambda: DeviceEnv("cartpole", seed=3), N, T, 0, 0.0, agent, seed=3)
    # every fourth env: pole at 0.205 rad falling at 3 rad/s -- the next step crosses 12 degrees whatever
    # the action -- and one step short of the time limit
    doomed = torch.arange(0, N, 4, device="cuda")
    ve.batch.state[doomed] = torch.tensor([0.0, 0.0, 0.205, 3.0], dtype=torch.float64, device="cuda")
    ve.batch.elapsed[doomed] = 499
    ve.t_states[doomed] = torch.tensor([0.0, 0.0, 0.205, 3.0], device="cuda")
    ve.rollout()
    d, tr = ve.done_flags, ve.trunc_flags
    assert bool((d[0, doomed] == 1).all()) and bool((tr[0, doomed] == 1).all())      # both flags, natively
    assert int(((d == 1) & (tr == 1)).sum()) == doomed.numel()
    with torch.no_grad():                                                             # bootstrap value recorded
        want = agent.get_values(ve._term_obs[0, doomed])
    assert torch.allclose(ve._ev_dense[0, doomed], want, rtol=1e-5, atol=2e-6)
    assert float(ve.states.detach()[1, doomed].abs().max()) <= 0.05 + 1e-6            # and the env was reset
    assert ve.end_states.shape[0] == int(((d + tr) > 0).sum())                        # one terminal observation per cell
    # dense GAE (what train_step runs) == the oracle's dense restatement of :476-486, bit for bit
    evz = torch.where(tr > 0, ve._ev_dense, torch.zeros_like(ve._ev_dense))
    adv = ops.gae_dense(ve.rewards, ve.values, ve._ev_dense, d, tr, 0.99, 0.95)
    r_np, v_np, d_np, tr_np = (x.cpu().numpy() for x in (ve.rewards, ve.values, d, tr))
    assert np.array_equal(adv.cpu().numpy(), cref.gae_dense(r_np, v_np, evz.cpu().numpy(), d_np, tr_np, 0.99, 0.95))
    # ... == the reference's ragged form with the kernel's own bootstrap values in the reference's order
    end_values = ve._ev_dense.t()[((d + tr) > 0).t()]                                 # actor-major, time-ascending (:401-402)
    end_values = torch.where((tr.t()[((d + tr) > 0).t()]) > 0, end_values, torch.zeros_like(end_values))
    ragged = ops.gae_ragged(ve.rewards, ve.values, end_values, d, tr, 0.99, 0.95)
    assert np.array_equal(ragged.cpu().numpy(), cref.gae_ragged(r_np, v_np, end_values.cpu().numpy(), d_np, tr_np, 0.99, 0.95))
    assert torch.equal(ragged, adv)
