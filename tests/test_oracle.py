"""CPU suite: the oracle (oracle/ppo_oracle.c, oracle/pyref.py) against the reference's own
known answers and the golden vectors produced by running the reference (tests/golden/,
oracle/gen_golden.py).  No GPU, no /root/reference at run time."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import cref, pyref


@pytest.fixture(scope="module")
def tables(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_test_tables.npz"))


# -- the literal tables printed in the reference's tests/test_ppo.py ---------------------------
TARGET_8x2 = np.array([[1.9285, 3.6539], [0.7979, 2.1966], [0.2296, 2.0272], [-0.8000, 0.0396],
                       [2.4389, -0.8000], [0.8090, 2.8596], [-0.3902, 2.4090], [-0.4000, 1.1940]], np.float32)
TARGET_20x1 = np.array([3.6492, 3.0486, 2.4078, 0.0296, -0.8000, 4.0158, 2.1624, 1.8973, 1.1880, 0.3157,
                        1.4160, 0.4593, -1.4000, 2.9224, -0.0570, 0.3860, 6.7316, 5.9028, 3.1077, 0.7780],
                       np.float32)[:, None]


def test_oracle_gae_reference_test_tables(tables):
    """reference tests/test_ppo.py:13-88 (test_ep_advantages), same tolerances"""
    a1 = cref.gae_ragged(tables["r1"], tables["v1"], tables["e1"], tables["d1"], tables["t1"], 0.99, 0.95)
    assert a1.shape == (8, 2) and a1.dtype == np.float32
    assert np.abs(a1 - TARGET_8x2).mean() < 1e-4
    assert np.array_equal(a1, tables["a1"])          # bit-exact vs the reference's full-precision output
    a2 = cref.gae_ragged(tables["r2"], tables["v2"], tables["e2"], tables["d2"], tables["t2"], 0.99, 0.95)
    assert np.abs(a2 - TARGET_20x1).mean() < 1e-3
    assert np.array_equal(a2, tables["a2"])


def test_oracle_gae_both_flags_probe(tables):
    """SURVEY.md 8c probe: done and trunc on the same step (reference src/ppo.py:476-480)"""
    r = np.ones((4, 1), np.float32)
    v = np.array([[1.0], [2.0], [3.0], [4.0]], np.float32)
    d = np.array([[0], [1], [0], [0]], np.int32)
    tr = np.array([[0], [1], [0], [1]], np.int32)
    a = cref.gae_ragged(r, v, np.array([5.0, 6.0], np.float32), d, tr, 0.99, 0.95)
    assert np.array_equal(a, tables["probe_both"])
    assert np.allclose(a.ravel(), [-1.2778246, -3.4639282, 4.7250700, 2.9400001], rtol=1e-6)


def test_oracle_gae_golden_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_gae.npz"))
    n = int(g["n_cases"])
    assert n >= 50
    for i in range(n):
        p = f"c{i}_"
        gamma, lam = g[p + "gl"]
        assert int(g[p + "err"]) == 0
        adv = cref.gae_ragged(g[p + "r"], g[p + "v"], g[p + "ev"], g[p + "done"], g[p + "trunc"], gamma, lam)
        assert np.array_equal(adv, g[p + "adv"]), f"case {i}"


def test_oracle_gae_dense_equals_ragged(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_gae.npz"))
    checked = 0
    for i in range(int(g["n_cases"])):
        p = f"c{i}_"
        d, tr = g[p + "done"], g[p + "trunc"]
        if ((d + tr) > 1).any() or ((d + tr) > 0).sum() == 0:
            continue  # both-flags skews the ragged index (reference quirk); dense has no such notion
        if ((d + tr)[-1] == 0).any():
            continue  # ragged form reads a wrapped (meaningless) bootstrap there; dense reads none
        gamma, lam = g[p + "gl"]
        dense = cref.ragged_to_dense(g[p + "ev"], d, tr)
        adv, ret = cref.gae_dense(g[p + "r"], g[p + "v"], dense, d, tr, gamma, lam, want_returns=True)
        assert np.array_equal(adv, g[p + "adv"]), f"case {i}"
        assert np.array_equal(ret, adv + g[p + "v"])
        checked += 1
    assert checked >= 15


def test_oracle_gae_index_error():
    r = np.ones((2, 1), np.float32)
    with pytest.raises(IndexError):  # trunc raised but end_values too short for the both-flags decrement
        cref.gae_ragged(r, r, np.array([1.0], np.float32), np.array([[1], [1]], np.int32),
                        np.array([[1], [1]], np.int32), 0.99, 0.95)


def test_oracle_gae_vs_torch_restatement():
    rng = np.random.default_rng(3)
    T, N = 37, 11
    r, v = rng.standard_normal((T, N)).astype(np.float32), rng.standard_normal((T, N)).astype(np.float32)
    d = (rng.random((T, N)) < 0.1).astype(np.int32)
    tr = ((rng.random((T, N)) < 0.05) & (d == 0)).astype(np.int32)
    tr[-1] = 1 - d[-1]
    ev = rng.standard_normal(int((d + tr).sum())).astype(np.float32)
    a = cref.gae_ragged(r, v, ev, d, tr, 0.99, 0.95)
    b = pyref.gae_torch(*(torch.tensor(x) for x in (r, v, ev, d, tr)), 0.99, 0.95).numpy()
    assert np.array_equal(a, b)


# -- loss ---------------------------------------------------------------------------------------
def test_oracle_loss_reference_test_table(tables):
    """reference tests/test_ppo.py:90-122 (test_compute_losses), same tolerances"""
    out4, _, _ = cref.loss(tables["ratios"], None, tables["cv"], tables["pv"], tables["ad"], 0.2, 1, 0, 1)
    for got, want in zip(out4, (0.0491, 1.5520, 0.4000, 0.0117)):
        assert abs(got - want) < 1e-3
    assert np.allclose(out4, tables["loss_clipva0"], rtol=1e-5)
    out4, _, _ = cref.loss(tables["ratios"], None, tables["cv"], tables["pv"], tables["ad"], 0.2, 1, 1, 1)
    assert np.allclose(out4, tables["loss_clipva1"], rtol=1e-5)  # value loss 1.6319997 with clipping


def _rel(a, b):
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-30))


def test_oracle_loss_golden(golden_dir):
    L = np.load(os.path.join(golden_dir, "ref_loss.npz"))
    n = int(L["n_cases"])
    assert n >= 40
    for i in range(n):
        p = f"c{i}_"
        norm_adv, clip_va, clip = L[p + "cfg"]
        args = (L[p + "new_v"], L[p + "old_v"], L[p + "adv"], clip, int(norm_adv), int(clip_va))
        out4, g_lp, g_v = cref.loss(L[p + "new_lp"], L[p + "old_lp"], *args, 0)
        ref = L[p + "out4"]
        assert np.allclose(out4[:3], ref[:3], rtol=1e-5, atol=1e-7), f"case {i}"
        assert abs(out4[3] - ref[3]) <= 1e-5 * abs(ref[3]) + 2e-7, f"case {i} kl"  # f32 cancellation in rho-1-log(rho)
        assert _rel(g_lp, L[p + "g_lp"]) < 1e-5 and _rel(g_v, L[p + "g_v"]) < 1e-5, f"case {i}"
        out4r, g_ratio, _ = cref.loss(L[p + "ratio"], None, *args, 1)
        assert np.allclose(out4r[:3], ref[:3], rtol=1e-5, atol=1e-7)
        assert _rel(g_ratio, L[p + "g_ratio"]) < 1e-5, f"case {i}"


def test_torch_restatement_of_compute_losses_golden(golden_dir):
    """oracle/pyref.losses_torch (bench.py's CPU loss baseline) against the reference-run goldens,
    values and autograd gradients."""
    import torch
    from oracle import pyref
    L = np.load(os.path.join(golden_dir, "ref_loss.npz"))
    for i in range(int(L["n_cases"])):
        p = f"c{i}_"
        norm_adv, clip_va, clip = L[p + "cfg"]
        ratio = torch.from_numpy(L[p + "ratio"]).requires_grad_(True)
        new_v = torch.from_numpy(L[p + "new_v"]).requires_grad_(True)
        pl, vl, cf, kl = pyref.losses_torch(ratio, new_v, torch.from_numpy(L[p + "old_v"]), torch.from_numpy(L[p + "adv"]),
                                            float(clip), int(norm_adv), int(clip_va))
        ref = L[p + "out4"]
        got = np.array([float(pl), float(vl), float(cf), float(kl)])
        if not np.isfinite(ref).all():                       # B = 1 with normalisation: NaN like the reference
            assert np.array_equal(np.isnan(got), np.isnan(ref))
            continue
        assert np.allclose(got, ref, rtol=1e-6, atol=1e-7), f"case {i}"
        (pl + vl).backward()
        assert _rel(ratio.grad.numpy(), L[p + "g_ratio"]) < 1e-5 and _rel(new_v.grad.numpy(), L[p + "g_v"]) < 1e-5


# -- Philox ---------------------------------------------------------------------------------------
def test_oracle_loss_logits_golden(golden_dir):
    """Discrete head restatement (Categorical log_prob / entropy + loss) vs the reference's
    Agent.get_actions_and_values + compute_losses outputs and autograd gradients."""
    L = np.load(os.path.join(golden_dir, "ref_loss_logits.npz"))
    for i in range(int(L["n_cases"])):
        p = f"c{i}_"
        norm_adv, clip_va, clip = L[p + "cfg"]
        out5, gpl, gent, gv = cref.loss_logits(L[p + "logits"], L[p + "actions"], L[p + "old_lp"], L[p + "new_v"],
                                               L[p + "old_v"], L[p + "adv"], float(clip), int(norm_adv), int(clip_va))
        ref = L[p + "out5"]
        assert np.allclose(out5[:3], ref[:3], rtol=2e-5, atol=1e-7), f"case {i}"
        assert abs(out5[3] - ref[3]) <= 1e-5 * abs(ref[3]) + 2e-7, f"case {i}"
        assert abs(out5[4] - ref[4]) <= 1e-5 * abs(ref[4]) + 1e-7, f"case {i}"
        for got, key in ((gpl, "g_pl"), (gent, "g_ent"), (gv, "g_v")):
            want = L[p + key]
            assert np.abs(got - want).max() <= 2e-5 * max(np.abs(want).max(), 1e-30) + 1e-9, f"case {i} {key}"


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32 10 rounds"""
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        assert cref.philox4x32_10(ctr, key).tolist() == want


def test_reset_uniforms_in_range_and_distinct():
    u = np.stack([cref.reset_uniform4(7, n, e) for n in range(200) for e in range(5)])
    assert (u >= 0).all() and (u < 1).all()
    assert len(np.unique(u)) == u.size
    assert abs(u.mean() - 0.5) < 0.03


# -- physics: hand-derived anchors; physics parity is UNPINNED ------------
def test_cartpole_first_step_known_answer():
    b = cref.BatchEnv("cartpole", 1)
    b.reset(reset_draw=np.full((1, 4), 0.5))  # U = 0.5 -> state exactly 0
    assert np.array_equal(b.state, np.zeros((1, 4)))
    obs, _, r, d, tr = b.step([1])
    assert np.allclose(b.state[0], [0.0, 0.02 * 9.75609756097561, 0.0, 0.02 * -14.634146341463415], rtol=1e-15)
    assert np.allclose(b.state[0], [0.0, 0.1951219512195122, 0.0, -0.2926829268292683], rtol=1e-14)
    assert r[0] == 1.0 and d[0] == 0 and tr[0] == 0
    assert obs.dtype == np.float32 and np.array_equal(obs[0], b.state[0].astype(np.float32))


def test_cartpole_thresholds_and_time_limit():
    b = cref.BatchEnv("cartpole", 2)
    b.reset(reset_draw=np.full((2, 4), 0.5))
    b.state[:] = [[0, 0, 0.2095, 0], [0, 0, 0.2093, 0]]  # theta_threshold = 0.20943951...
    _, term_obs, _, d, _ = b.step([1, 1])
    assert d.tolist() == [1, 0]
    assert term_obs[0, 2] > 0.2094                       # terminal obs captured, env 0 was reset
    assert abs(b.state[0, 2]) <= 0.05 and b.elapsed[0] == 0
    # time limit: 499 elapsed steps + 1 => truncated (also when not terminated)
    b = cref.BatchEnv("cartpole", 1)
    b.reset(reset_draw=np.full((1, 4), 0.5))
    b.elapsed[:] = 499
    _, _, _, d, tr = b.step([0])
    assert d[0] == 0 and tr[0] == 1 and b.elapsed[0] == 0 and b.ring_cnt[0] == 1 and b.len_ring[0, 0] == 1


def test_forced_truncation_keeps_env_running():
    b = cref.BatchEnv("cartpole", 1)
    b.reset(reset_draw=np.full((1, 4), 0.5))
    obs, term_obs, _, d, tr = b.step([1], is_last=True)
    assert d[0] == 0 and tr[0] == 1                      # reference src/ppo.py:353-356
    assert b.elapsed[0] == 1 and b.ep_len[0] == 1 and b.ring_cnt[0] == 0  # no reset, no episode recorded
    assert np.array_equal(obs, term_obs)                 # end state == carried state


def test_acrobot_rest_is_fixed_point_and_terminal_height():
    b = cref.BatchEnv("acrobot", 1)
    b.reset(reset_draw=np.full((1, 4), 0.5))             # hanging at rest, theta = 0
    for _ in range(5):
        obs, _, r, d, tr = b.step([1])                   # torque 0
    assert np.allclose(b.state, 0.0, atol=1e-15) and r[0] == -1.0 and d[0] == 0
    assert np.allclose(obs[0], [1, 0, 1, 0, 0, 0], atol=1e-7)
    # nearly upright first link: -cos(t1) - cos(t1+t2) = 2 > 1 => terminated, reward 0
    b.state[:] = [[math.pi - 1e-3, 0.0, 0.0, 0.0]]
    _, _, r, d, _ = b.step([1])
    assert d[0] == 1 and r[0] == 0.0


@pytest.mark.parametrize("kind", ["cartpole", "acrobot"])
def test_c_oracle_equals_python_restatement(kind):
    """ppo_oracle.c and the pure-Python per-env restatement agree bit for bit (same host libm)."""
    rng = np.random.default_rng(11)
    N, steps = 6, 700
    b = cref.BatchEnv(kind, N)
    envs = [pyref.make_env(kind) for _ in range(N)]
    u0 = rng.random((N, 4))
    obs_c = b.reset(reset_draw=u0)
    obs_p = np.stack([e.reset(u=u0[i])[0] for i, e in enumerate(envs)])
    assert np.allclose(obs_c, obs_p, rtol=0, atol=1e-7)
    na = 2 if kind == "cartpole" else 3
    for t in range(steps):
        a = rng.integers(0, na, N)
        if kind == "cartpole" and t > 100:  # keep some poles up long enough to hit the time limit
            a = (b.state[:, 2] + 0.5 * b.state[:, 3] > 0).astype(np.int64)
        u = rng.random((N, 4))
        obs_c, term_c, r_c, d_c, tr_c = b.step(a, reset_draw=u)
        for i, e in enumerate(envs):
            o, r, d, tr, _ = e.step(int(a[i]))
            assert (int(d), int(tr)) == (d_c[i], tr_c[i]) and np.float32(r) == r_c[i]
            if d or tr:
                assert np.array_equal(o, term_c[i])
                o = e.reset(u=u[i])[0]
                assert np.allclose(o, obs_c[i], rtol=0, atol=1e-7)  # f32-computed cos/sin right after reset
            else:
                assert np.array_equal(o, obs_c[i])
            assert np.array_equal(np.asarray(e.env.state, dtype=np.float64), b.state[i])
    assert b.ring_cnt.sum() > 0
    for i, e in enumerate(envs):
        k = min(int(b.ring_cnt[i]), 100)
        assert sorted(np.array(e.length_queue).tolist()) == sorted(b.len_ring[i, :k].tolist())


def test_three_fma_division_by_total_mass_is_ieee_division():
    """The HIP CartPole step divides by total_mass = 1.1 with q = x r, q' = fma(fma(-q, 1.1, x), r, q)
    (env_core.h div_by_total_mass; Markstein's correction with a correctly rounded reciprocal).  The
    reference divides (gymnasium cartpole.py, three `/ self.total_mass`); the claim is equality, checked
    here on 2e8 random doubles over the binades the physics can reach and on both ends of the double range
    short of overflow / subnormals (1.5e9 samples over [2^-40, 2^20) were checked once offline)."""
    assert cref.div_by_total_mass_mismatches(150_000_000, seed=1, e_lo=-40, e_hi=20) == 0
    assert cref.div_by_total_mass_mismatches(25_000_000, seed=2, e_lo=-900, e_hi=-40) == 0
    assert cref.div_by_total_mass_mismatches(25_000_000, seed=3, e_lo=20, e_hi=1000) == 0


def test_shuffle_restatement_is_a_uniform_looking_permutation():
    """oracle_shuffle_indices (what ppo_shuffle_indices computes; tests/test_gpu_shuffle.py compares them bit for bit):
    a permutation for every n; over many (seed, stream) pairs every element lands on every position equally often
    (chi-square), neighbours do not stay neighbours, fixed points are Poisson(1), and input / output positions are
    uncorrelated."""
    from oracle import cref
    for n in (1, 2, 3, 7, 8, 9, 100, 1023, 1024, 1025, 50000):
        p = cref.shuffle_indices(n, 5, n)
        assert np.array_equal(np.sort(p), np.arange(n)), n
    # position uniformity, small n: counts[i, j] = how often element j sits at position i
    trials = 4000
    for n in (5, 8, 13):
        counts = np.zeros((n, n))
        for s in range(trials):
            p = cref.shuffle_indices(n, 1000 + s, s // 7)
            counts[np.arange(n), p] += 1
        chi2 = ((counts - trials / n) ** 2 / (trials / n)).sum()      # (n - 1)^2 degrees of freedom
        dof = (n - 1) ** 2
        assert chi2 < dof + 5 * np.sqrt(2 * dof), (n, chi2, dof)
    # a large permutation: correlation, adjacency, fixed points (averaged over streams)
    n = 1 << 16
    fixed, adjacent = [], []
    for s in range(40):
        p = cref.shuffle_indices(n, 77, s)
        r = np.corrcoef(np.arange(n), p)[0, 1]
        assert abs(r) < 5 / np.sqrt(n), (s, r)
        fixed.append(int((p == np.arange(n)).sum()))
        adjacent.append(int((np.abs(np.diff(p)) == 1).sum()))          # expected ~ 2
    assert 0.4 < np.mean(fixed) < 1.8, np.mean(fixed)
    assert np.mean(adjacent) < 4.0, np.mean(adjacent)
    # a non-power-of-two size close above a power of two (cycle walking nearly every other index)
    n = (1 << 15) + 3
    p = cref.shuffle_indices(n, 1, 2)
    assert np.array_equal(np.sort(p), np.arange(n)) and abs(np.corrcoef(np.arange(n), p)[0, 1]) < 0.03
