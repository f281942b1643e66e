"""GPU parity: ppo_env_reset / ppo_env_step (through the C ABI) vs the CPU oracle
(oracle/ppo_oracle.c, physics PARITY UNPINNED at the gymnasium boundary -- see its header).

Bar: done / trunc flags, TimeLimit counters, episode lengths and reset counters BIT-EXACT;
fp64 state and f32 observations within 1e-5 relative (measured: a few ulp -- only the device
sin/cos differ from glibc's).  Philox reset draws are integer arithmetic: bit-exact."""
import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu
RTOL = 1e-5


class DevBatch:
    """Thin driver around ppo_amd.envs.DeviceEnvBatch mirroring oracle.cref.BatchEnv.step."""

    def __init__(self, kind, N, seed=0, env_base=0):
        from ppo_amd.envs import DeviceEnvBatch
        self.b = DeviceEnvBatch(kind, N, seed=seed, env_base=env_base)
        d, od = self.b.device, self.b.obs_dim
        self.obs = torch.empty((N, od), dtype=torch.float32, device=d)
        self.term = torch.zeros((N, od), dtype=torch.float32, device=d)
        self.rew = torch.empty(N, dtype=torch.float32, device=d)
        self.done = torch.empty(N, dtype=torch.int32, device=d)
        self.trunc = torch.empty(N, dtype=torch.int32, device=d)

    def reset(self, reset_draw=None):
        rd = None if reset_draw is None else torch.from_numpy(np.asarray(reset_draw, np.float64)).cuda()
        return self.b.reset(self.obs, rd).cpu().numpy()

    def step(self, action, is_last=False, reset_draw=None):
        a = torch.from_numpy(np.asarray(action, np.int32)).cuda()
        rd = None if reset_draw is None else torch.from_numpy(np.asarray(reset_draw, np.float64)).cuda()
        self.term.zero_()
        self.b.step(a, self.obs, self.term, self.rew, self.done, self.trunc, is_last=is_last, reset_draw=rd)
        return tuple(x.cpu().numpy() for x in (self.obs, self.term, self.rew, self.done, self.trunc))


def sync_state(dev: DevBatch, ora: cref.BatchEnv):
    """Copy the oracle's state into the device batch (identical inputs for a single-step check)."""
    dev.b.state.copy_(torch.from_numpy(ora.state))
    dev.b.elapsed.copy_(torch.from_numpy(ora.elapsed))
    dev.b.ep_ret.copy_(torch.from_numpy(ora.ep_ret))
    dev.b.ep_len.copy_(torch.from_numpy(ora.ep_len))


@pytest.mark.parametrize("kind", ["cartpole", "acrobot"])
def test_reset_philox_bit_exact(kind):
    N = 5000
    dev, ora = DevBatch(kind, N, seed=1234, env_base=10**10), cref.BatchEnv(kind, N, seed=1234, env_base=10**10)
    for _ in range(3):  # successive resets walk the per-env episode counter
        o_d, o_c = dev.reset(), ora.reset()
        assert np.array_equal(dev.b.state.cpu().numpy(), ora.state)          # fp64, bit-exact
        assert np.allclose(o_d, o_c, rtol=0, atol=6e-8)                      # acrobot: f32 cos/sin of f32 angle
        assert np.array_equal(dev.b.episode.cpu().numpy().view(np.uint32), ora.episode)
    lim = 0.05 if kind == "cartpole" else 0.1
    s = ora.state
    assert (np.abs(s) <= lim).all() and abs(s.mean()) < lim / 20 and abs(s.std() - 2 * lim / 12**0.5) < lim / 20


def test_cartpole_known_answer_first_step():
    dev = DevBatch("cartpole", 1)
    dev.reset(np.full((1, 4), 0.5))
    obs, _, r, d, tr = dev.step([1])
    st = dev.b.state.cpu().numpy()[0]
    assert np.allclose(st, [0.0, 0.1951219512195122, 0.0, -0.2926829268292683], rtol=1e-14)
    assert r[0] == 1.0 and d[0] == 0 and tr[0] == 0 and np.array_equal(obs[0], st.astype(np.float32))


@pytest.mark.parametrize("kind,na", [("cartpole", 2), ("acrobot", 3)])
def test_single_step_from_identical_states(kind, na):
    """1M independent states covering the whole reachable box, incl. both sides of every threshold."""
    N = 1 << 20
    rng = np.random.default_rng(42)
    dev, ora = DevBatch(kind, N, seed=5), cref.BatchEnv(kind, N, seed=5)
    dev.reset(); ora.reset()
    if kind == "cartpole":
        ora.state[:] = rng.uniform(-1, 1, (N, 4)) * [2.6, 3.0, 0.23, 3.5]
    else:
        ora.state[:] = rng.uniform(-1, 1, (N, 4)) * [np.pi, np.pi, 4 * np.pi, 9 * np.pi]
    ora.elapsed[:] = rng.integers(0, 500, N)
    ora.elapsed[:1000] = 499
    sync_state(dev, ora)
    a = rng.integers(0, na, N)
    u = rng.random((N, 4))
    o_d, t_d, r_d, d_d, tr_d = dev.step(a, reset_draw=u)
    o_c, t_c, r_c, d_c, tr_c = ora.step(a, reset_draw=u)
    assert np.array_equal(d_d, d_c) and np.array_equal(tr_d, tr_c)           # flags bit-exact
    assert d_c.sum() > 1000 and tr_c.sum() >= 1000 and ((d_c + tr_c) == 2).sum() > 0
    assert np.array_equal(r_d, r_c)
    assert np.array_equal(dev.b.elapsed.cpu().numpy(), ora.elapsed)
    assert np.array_equal(dev.b.ring_cnt.cpu().numpy(), ora.ring_cnt)
    assert np.array_equal(dev.b.len_ring.cpu().numpy(), ora.len_ring)
    st_d = dev.b.state.cpu().numpy()
    err = np.abs(st_d - ora.state) / np.maximum(np.abs(ora.state), 1.0)
    # CartPole: one Euler step, error stays at the ulp level (measured 2.2e-16).  Acrobot: the box
    # corners (|w1| ~ 4pi with |w2| ~ 9pi) drive RK4 stage accelerations to ~1e6 before the velocity
    # clip, so an ulp in a stage cos/sin is amplified to ~1e-12 absolute (measured 3.4e-12).
    assert err.max() < (1e-14 if kind == "cartpole" else 1e-10), err.max()
    assert np.allclose(o_d, o_c, rtol=RTOL, atol=1e-7) and np.allclose(t_d, t_c, rtol=RTOL, atol=1e-7)
    frac_bitexact = (st_d == ora.state).mean()
    print(f"{kind}: fp64 state bit-identical to glibc-libm oracle for {100 * frac_bitexact:.3f}% of values; "
          f"max rel err {err.max():.2e}")


def test_cartpole_step_from_a_state_past_the_termination_band():
    """A pole angle that no transition leaves behind (|theta| up to 3 rad; gymnasium only warns when an env is stepped
    past its terminal state): the kernels take the general sin / cos for the step after the load
    (csrc/env_core.h CartPole::Aux).  Per-step kernel and the whole-rollout kernel's first step, whole waves of
    such states and waves where only some lanes hold one."""
    from ppo_amd import Agent, SyncVecEnv
    from ppo_amd.envs import DeviceEnv
    N = 4096
    rng = np.random.default_rng(7)
    dev, ora = DevBatch("cartpole", N, seed=5), cref.BatchEnv("cartpole", N, seed=5)
    dev.reset(); ora.reset()
    st = rng.uniform(-1, 1, (N, 4)) * [1.0, 2.0, 3.0, 2.0]
    st[1024:, 2] *= 0.05                             # waves inside the band
    st[2048::97, 2] = 1.7                            # single lanes outside it
    ora.state[:] = st
    sync_state(dev, ora)
    a = rng.integers(0, 2, N)
    u = rng.random((N, 4))
    o_d, t_d, r_d, d_d, tr_d = dev.step(a, reset_draw=u)
    o_c, t_c, r_c, d_c, tr_c = ora.step(a, reset_draw=u)
    assert np.array_equal(d_d, d_c) and np.array_equal(tr_d, tr_c) and 900 < d_c.sum() < N
    assert np.allclose(t_d, t_c, rtol=RTOL, atol=1e-7) and np.allclose(o_d, o_c, rtol=RTOL, atol=1e-7)
    assert np.abs(dev.b.state.cpu().numpy() - ora.state).max() < 1e-14

    # the whole-rollout kernel: same states, its own sampled actions replayed through the oracle for one step
    torch.manual_seed(0)
    probe = DeviceEnv("cartpole")
    agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
    for n_envs in (N, 1000):                         # the form without the per-lane branch, and the general one
        ve = SyncVecEnv(lambda: DeviceEnv("cartpole", seed=5), n_envs, 4, 0, 0.0, agent, seed=5)
        ve.wave_envs = 64
        ora = cref.BatchEnv("cartpole", n_envs, seed=5)
        ora.reset()
        ora.state[:] = st[:n_envs]
        ve.batch.state.copy_(torch.from_numpy(ora.state))
        ve.t_states.copy_(torch.from_numpy(ora.state.astype(np.float32)))
        ve.rollout()
        o_c, t_c, r_c, d_c, tr_c = ora.step(ve.actions[0].cpu().numpy())
        assert np.array_equal(ve.done_flags[0].cpu().numpy(), d_c) and d_c.sum() > 200
        assert np.allclose(ve.states[1].detach().cpu().numpy(), o_c, rtol=RTOL, atol=1e-7)
        ended = torch.from_numpy(d_c > 0)
        assert np.allclose(ve._term_obs[0].cpu()[ended].numpy(), t_c[d_c > 0], rtol=RTOL, atol=1e-7)


def test_acrobot_step_from_a_state_outside_the_clip_bounds():
    """Angles beyond +-pi and velocities beyond the clip bounds -- states no transition leaves behind, only a caller can
    write them: the kernels run the GENERAL form of the transition for such a wave (libm's sincos instead of the
    bounded-range one, csrc/env_core.h Acrobot TWO_FORMS).  Per-step kernel and the whole-rollout kernel's first step;
    waves made of such states, waves with single lanes of them, and plain waves beside them."""
    from ppo_amd import Agent, SyncVecEnv
    from ppo_amd.envs import DeviceEnv
    N = 4096
    rng = np.random.default_rng(9)
    dev, ora = DevBatch("acrobot", N, seed=5), cref.BatchEnv("acrobot", N, seed=5)
    dev.reset(); ora.reset()
    st = rng.uniform(-1, 1, (N, 4)) * [np.pi, np.pi, 4 * np.pi, 9 * np.pi]
    st[:1024] *= [40.0, 40.0, 3.0, 3.0]              # whole waves outside: angles to 125 rad, velocities to 3x the clip
    st[2048::97, 3] = 100.0                          # single lanes outside
    st[2048::89, 0] = -7.0
    ora.state[:] = st
    sync_state(dev, ora)
    a = rng.integers(0, 3, N)
    u = rng.random((N, 4))
    o_d, t_d, r_d, d_d, tr_d = dev.step(a, reset_draw=u)
    o_c, t_c, r_c, d_c, tr_c = ora.step(a, reset_draw=u)
    assert np.array_equal(d_d, d_c) and np.array_equal(tr_d, tr_c) and d_c.sum() > 100
    assert np.allclose(t_d, t_c, rtol=RTOL, atol=1e-7) and np.allclose(o_d, o_c, rtol=RTOL, atol=1e-7)
    err = np.abs(dev.b.state.cpu().numpy() - ora.state) / np.maximum(np.abs(ora.state), 1.0)
    assert err.max() < 1e-9, err.max()               # (stage accelerations of ~1e7 amplify an ulp of a stage's sin / cos)

    torch.manual_seed(0)
    probe = DeviceEnv("acrobot")
    agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
    for n_envs in (N, 1000):                         # 64-env waves without / with the per-lane branch
        for width in (64, 32):
            ve = SyncVecEnv(lambda: DeviceEnv("acrobot", seed=5), n_envs, 4, 0, 0.0, agent, seed=5)
            ve.wave_envs = width
            ora = cref.BatchEnv("acrobot", n_envs, seed=5)
            ora.reset()
            ora.state[:] = st[:n_envs]
            ve.batch.state.copy_(torch.from_numpy(ora.state))
            obs0 = np.stack([np.cos(st[:n_envs, 0]), np.sin(st[:n_envs, 0]), np.cos(st[:n_envs, 1]), np.sin(st[:n_envs, 1]),
                             st[:n_envs, 2], st[:n_envs, 3]], 1).astype(np.float32)
            ve.t_states.copy_(torch.from_numpy(obs0))
            ve.rollout()
            o_c, t_c, r_c, d_c, tr_c = ora.step(ve.actions[0].cpu().numpy())
            assert np.array_equal(ve.done_flags[0].cpu().numpy(), d_c) and d_c.sum() > 20
            assert np.allclose(ve.states[1].detach().cpu().numpy(), o_c, rtol=RTOL, atol=1e-7)
            ended = torch.from_numpy(d_c > 0)
            assert np.allclose(ve._term_obs[0].cpu()[ended].numpy(), t_c[d_c > 0], rtol=RTOL, atol=1e-7)


@pytest.mark.parametrize("kind,na,steps", [("cartpole", 2, 1100), ("acrobot", 3, 1100)])
def test_trajectories_flags_and_episode_lengths_bit_exact(kind, na, steps):
    """2048 envs x 1100 steps along realistic trajectories (random policy + a balancing controller on
    half the CartPoles so that TimeLimit truncations occur), same injected actions and reset draws.
    Every step: flags, rewards bit-exact, observations to tolerance.  All integer bookkeeping
    (TimeLimit counters, episode lengths/returns rings, reset counters) is carried independently on
    both sides for the whole run and must be identical at the end.
    The fp64 physics state is copied oracle -> device before each step: both systems are chaotic
    (open-loop replay of actions on an inverted pendulum amplifies one ulp of sin/cos difference by
    ~e^39 over a 500-step episode), so free-running trajectories of ANY two libm implementations
    decorrelate; what is comparable is every single transition along the way."""
    N = 2048
    rng = np.random.default_rng(7)
    dev, ora = DevBatch(kind, N, seed=99), cref.BatchEnv(kind, N, seed=99)
    u0 = rng.random((N, 4))
    assert np.allclose(dev.reset(u0), ora.reset(u0), atol=1e-7)
    n_done = n_trunc = 0
    worst = 0.0
    for t in range(steps):
        a = rng.integers(0, na, N)
        if kind == "cartpole":  # PD controller on the first half so episodes reach 500 steps
            ctl = (ora.state[:, 2] + 0.5 * ora.state[:, 3] + 0.05 * ora.state[:, 0] + 0.1 * ora.state[:, 1] > 0)
            a[: N // 2] = ctl[: N // 2]
        u = rng.random((N, 4))
        last = (t % 128 == 127)
        dev.b.state.copy_(torch.from_numpy(ora.state))
        o_d, t_d, r_d, d_d, tr_d = dev.step(a, is_last=last, reset_draw=u)
        o_c, t_c, r_c, d_c, tr_c = ora.step(a, is_last=last, reset_draw=u)
        assert np.array_equal(d_d, d_c) and np.array_equal(tr_d, tr_c), f"flags differ at step {t}"
        assert np.array_equal(r_d, r_c)
        assert np.allclose(o_d, o_c, rtol=RTOL, atol=1e-7), f"obs differ at step {t}"
        assert np.allclose(t_d, t_c, rtol=RTOL, atol=1e-7)
        worst = max(worst, float(np.abs(dev.b.state.cpu().numpy() - ora.state).max()))
        n_done += int(d_c.sum())
        if not last:
            n_trunc += int(tr_c.sum())
    if kind == "cartpole":
        assert n_done > 1000 and n_trunc > 100  # terminations and real TimeLimit truncations happened
    else:
        assert n_done > 20 and n_trunc > 2000   # a random policy rarely swings Acrobot up
    assert np.array_equal(dev.b.elapsed.cpu().numpy(), ora.elapsed)
    assert np.array_equal(dev.b.ep_len.cpu().numpy(), ora.ep_len)
    assert np.array_equal(dev.b.ep_ret.cpu().numpy(), ora.ep_ret)
    assert np.array_equal(dev.b.ring_cnt.cpu().numpy(), ora.ring_cnt)
    assert np.array_equal(dev.b.len_ring.cpu().numpy(), ora.len_ring)        # episode lengths bit-exact
    assert np.array_equal(dev.b.ret_ring.cpu().numpy(), ora.ret_ring)
    assert np.array_equal(dev.b.episode.cpu().numpy().view(np.uint32), ora.episode)
    print(f"{kind}: worst |state_dev - state_oracle| after one step over {steps} steps x {N} envs: {worst:.2e}")


@pytest.mark.parametrize("kind", ["cartpole", "acrobot"])
def test_native_philox_resets_match_oracle_stream(kind):
    """No injected draws: device and oracle consume the same Philox stream keyed by global env id."""
    N, base = 777, 123456789012
    rng = np.random.default_rng(1)
    dev, ora = DevBatch(kind, N, seed=31337, env_base=base), cref.BatchEnv(kind, N, seed=31337, env_base=base)
    dev.reset(); ora.reset()
    ora.elapsed[:] = rng.integers(300, 499, N)      # so that every env is reset at least once (TimeLimit)
    dev.b.elapsed.copy_(torch.from_numpy(ora.elapsed))
    na = 2 if kind == "cartpole" else 3
    for t in range(300):
        a = rng.integers(0, na, N)
        dev.b.state.copy_(torch.from_numpy(ora.state))
        o_d, _, _, d_d, tr_d = dev.step(a)
        o_c, _, _, d_c, tr_c = ora.step(a)
        assert np.array_equal(d_d, d_c) and np.array_equal(tr_d, tr_c)
        assert np.allclose(o_d, o_c, rtol=RTOL, atol=1e-7)
    assert ora.episode.min() >= 2
    assert np.array_equal(dev.b.episode.cpu().numpy().view(np.uint32), ora.episode)
    # sharding invariance: the second half as its own batch with env_base shifted gives the same resets
    half = DevBatch(kind, N - 400, seed=31337, env_base=base + 400)
    full = DevBatch(kind, N, seed=31337, env_base=base)
    assert np.array_equal(half.reset(), full.reset()[400:])


def test_forced_truncation_and_empty_batch():
    dev = DevBatch("cartpole", 3)
    dev.reset(np.full((3, 4), 0.5))
    obs, term, _, d, tr = dev.step([1, 0, 1], is_last=True)
    assert d.tolist() == [0, 0, 0] and tr.tolist() == [1, 1, 1]
    assert np.array_equal(obs, term) and dev.b.elapsed.tolist() == [1, 1, 1] and dev.b.ring_cnt.sum().item() == 0
    from ppo_amd import _lib
    assert _lib.load().ppo_env_step(0, None, None, None, None, None, None, None, 0, 0, None, None, None, None,
                                    None, None, None, None, 0, 0, 0, None) == 0  # N = 0 is a no-op


def test_categorical_sample_distribution_and_logp():
    from ppo_amd import ops
    N, A = 1 << 18, 3
    logits = torch.tensor([[0.3, -1.2, 0.9]]).repeat(N, 1).cuda()
    a, lp = ops.categorical_sample(logits, seed=11, env_base=0, step=5)
    assert a.dtype == torch.int32 and lp.dtype == torch.float32 and int(a.min()) >= 0 and int(a.max()) < A
    want_lp = torch.log_softmax(logits, dim=1)
    assert torch.allclose(lp, want_lp.gather(1, a.long()[:, None])[:, 0], rtol=1e-5, atol=1e-6)
    freq = torch.bincount(a.long(), minlength=A).double() / N
    p = torch.softmax(logits[0].double(), 0)
    assert (freq - p).abs().max() < 5 * (0.25 / N) ** 0.5                     # 5 sigma
    a2, _ = ops.categorical_sample(logits, seed=11, env_base=0, step=5)
    a3, _ = ops.categorical_sample(logits, seed=11, env_base=0, step=6)
    assert torch.equal(a, a2) and not torch.equal(a, a3)


# ---- Acrobot: the termination flag at the threshold, and the RK4 stages' division --------------------------
def _states_at_the_threshold(n, width, seed):
    """float64 states [n, 4] whose -cos(theta1) - cos(theta1 + theta2) lies within `width` of 1.0 (both sides):
    theta1 in the upper half plane (either sign), theta2 solved in extended precision."""
    rng = np.random.default_rng(seed)
    ld = np.longdouble
    th1 = rng.uniform(np.pi / 2 + 0.05, np.pi - 0.05, n).astype(ld)
    delta = rng.uniform(-width, width, n).astype(ld)
    c = -ld(1) - delta - np.cos(th1)                      # cos(theta1 + theta2) that puts the lhs at 1 + delta
    th2 = np.arccos(c) - th1
    sign = np.where(rng.random(n) < 0.5, -1.0, 1.0)       # (theta1, theta2) -> (-theta1, -theta2): same cosines
    st = np.empty((n, 4), dtype=np.float64)
    st[:, 0], st[:, 1] = (sign * th1).astype(np.float64), (sign * th2).astype(np.float64)
    st[:, 2], st[:, 3] = rng.uniform(-4 * np.pi, 4 * np.pi, n), rng.uniform(-9 * np.pi, 9 * np.pi, n)
    return st


@pytest.mark.parametrize("width", [1e-12, 1e-6], ids=["inside_the_guard_band", "outside_it"])
def test_acrobot_termination_flag_is_the_reference_expressions(width):
    """gymnasium decides termination with -cos(s0) - cos(s1 + s0) > 1.0 (AcrobotEnv._terminal; call site reference
    src/env_acrobot.py:7).  The kernels take cos(theta1 + theta2) from the observation's sin / cos (angle-sum
    identity) but let that value decide only outside a band of 1e-9 around the threshold; inside it they evaluate the
    reference's expression itself.  A million states within 1e-12 of the threshold (and a million 1e-6 away): the
    flag equals the oracle's (glibc cos) everywhere except where the left-hand side IS the threshold to within the
    last bits -- there the device's and glibc's cos may differ by an ulp, which no formulation can remove."""
    from ppo_amd import _lib, ops
    lib = _lib.load()
    n = 1 << 20
    st = _states_at_the_threshold(n, width, seed=5)
    want, lhs_ref = cref.acrobot_terminal(st)
    assert np.all(np.abs(lhs_ref - 1.0) <= 1.001 * width + 1e-15)            # the construction put them there
    assert 0.4 < want.mean() < 0.6                                            # both sides of the threshold
    d_st = torch.from_numpy(st).cuda()
    flag = torch.empty(n, dtype=torch.int32, device="cuda")
    lhs = torch.empty(n, dtype=torch.float64, device="cuda")
    _lib.check(lib.ppo_test_acrobot_terminal(d_st.data_ptr(), flag.data_ptr(), lhs.data_ptr(), n, ops._stream()),
               "ppo_test_acrobot_terminal")
    got, lhs = flag.cpu().numpy(), lhs.cpu().numpy()
    assert np.abs(lhs - lhs_ref).max() < 2e-15                                # the identity's value: a few ulp from the direct one
    differ = got != want
    # only ON the threshold (|lhs - 1| within two ulp of 1.0) may the two libms disagree, and rarely
    assert np.all(np.abs(lhs_ref[differ] - 1.0) <= 4.5e-16), (differ.sum(), np.abs(lhs_ref[differ] - 1.0).max())
    assert differ.sum() <= (64 if width < 1e-9 else 0), differ.sum()
    print(f"acrobot termination, width {width:g}: {differ.sum()} of {n} flags differ from glibc's (all with |lhs - 1| <= 2 ulp)")


def test_acrobot_stage_division_equals_ieee_division():
    """The four divisions of an RK4 stage (gymnasium AcrobotEnv._dsdt: d2 / d1, d2**2 / d1, the quotient that gives
    ddtheta2, and -(...) / d1) run as Acrobot::quot -- hardware reciprocal, two Newton steps, one Markstein correction.
    Against IEEE division (the compiler's own sequence on the device, and numpy on the host for the first chunk) on
    2 * 10^8 operand pairs over the ranges the dynamics produce: denominators d1 = 3.5 + cos(theta2) in [2.5, 4.5]
    and 1.25 - d2^2 / d1 in [0.57, 1.03], numerators of either sign over twelve decades."""
    from ppo_amd import _lib, ops
    lib = _lib.load()
    chunk, chunks = 1 << 24, 12
    gen = torch.Generator(device="cuda").manual_seed(77)
    total = bad = 0
    for c in range(chunks):
        u = torch.rand(chunk, dtype=torch.float64, device="cuda", generator=gen)
        den = torch.where(torch.rand(chunk, device="cuda", generator=gen) < 0.5, 2.5 + 2.0 * u, 0.57 + 0.46 * u)
        mag = 10.0 ** (torch.rand(chunk, dtype=torch.float64, device="cuda", generator=gen) * 12.0 - 8.0)
        num = torch.where(torch.rand(chunk, device="cuda", generator=gen) < 0.5, -mag, mag)
        if c == 0:                                          # the operands the dynamics divide most often
            num[: chunk // 4] = 1.25 + 0.5 * torch.cos(u[: chunk // 4] * 6.283185307179586)     # d2
            num[chunk // 4: chunk // 2] = num[: chunk // 4] ** 2                                    # d2^2
        qf, qi = torch.empty_like(num), torch.empty_like(num)
        _lib.check(lib.ppo_test_acrobot_quot(num.data_ptr(), den.data_ptr(), qf.data_ptr(), qi.data_ptr(), chunk, ops._stream()),
                   "ppo_test_acrobot_quot")
        bad += int((qf.view(torch.int64) != qi.view(torch.int64)).sum())
        total += chunk
        if c == 0:
            host = num.cpu().numpy() / den.cpu().numpy()
            assert np.array_equal(host.view(np.int64), qi.cpu().numpy().view(np.int64))     # the device's `/` is IEEE
    assert total >= 2 * 10 ** 8 and bad == 0, (bad, total)
