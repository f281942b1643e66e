"""GPU parity: ppo_gae_dense / ppo_gae_ragged (through the C ABI) vs the oracle and the
reference-generated golden vectors.  Bar: BIT-EXACT (same f32 operation sequence)."""
import os

import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu

TARGET_8x2 = torch.tensor([[1.9285, 3.6539], [0.7979, 2.1966], [0.2296, 2.0272], [-0.8000, 0.0396],
                           [2.4389, -0.8000], [0.8090, 2.8596], [-0.3902, 2.4090], [-0.4000, 1.1940]])
TARGET_20x1 = torch.tensor([3.6492, 3.0486, 2.4078, 0.0296, -0.8000, 4.0158, 2.1624, 1.8973, 1.1880, 0.3157,
                            1.4160, 0.4593, -1.4000, 2.9224, -0.0570, 0.3860, 6.7316, 5.9028, 3.1077, 0.7780])[:, None]


def make_ppo(**kw):
    from ppo_amd import PPO
    args = dict(discount_factor=0.99, gae_factor=0.95, norm_adv=1, clip_va_loss=0, conv_net=0, joint_network=1,
                use_gpu=False)
    args.update(kw)
    return PPO(**args)


def test_ep_advantages():
    """Mirror of the reference's tests/test_ppo.py:13-88, same inputs, same tolerances."""
    ppo = make_ppo()
    rewards = torch.tensor([[1.0, 1.0], [1.0, 0.5], [1.0, 2.0], [1.0, 1.0], [1.0, 0.0], [1.0, 1.0], [1.0, 0.5], [1.0, 2.0]])
    values = torch.tensor([[2.0, 0.6], [2.2, 1.2], [1.8, 1.0], [1.8, 1.0], [1.5, 0.8], [2.2, 1.0], [2.4, 0.6], [1.4, 1.4]])
    end_values = torch.tensor([1.6, 1.8, 0.6])
    done_flags = torch.tensor([[0, 0], [0, 0], [0, 0], [1, 0], [0, 1], [0, 0], [0, 0], [0, 0]], dtype=torch.int32)
    trunc_flags = torch.tensor([[0, 0], [0, 0], [0, 0], [0, 0], [0, 0], [0, 0], [0, 0], [0, 1]], dtype=torch.int32)
    advantages = ppo.compute_advantages(rewards, values, end_values, done_flags, trunc_flags)
    assert advantages.size(0) == 8 and advantages.size(1) == 2
    assert advantages.dtype == torch.float32
    assert (advantages - TARGET_8x2).abs().mean() < 1e-4

    rewards = torch.tensor([1.0, 1.0, 2.0, 1.0, 1.0, 2.0, 0.0, 1.0, 2.0, 0.0, 1.0, 1.0, 1.0, 3.0, 0.0, 1.0, 1.0, 3.0, 2.0, 1.0]).unsqueeze(1)
    values = torch.tensor([2.0, 1.8, 1.6, 2.0, 1.8, 1.8, 1.8, 2.2, 2.0, 2.6, 1.6, 1.6, 2.4, 2.4, 2.4, 2.0, 1.8, 2.0, 2.0, 2.4]).unsqueeze(1)
    end_values = torch.tensor([2.4, 1.2, 1.8, 1.4, 2.2])
    done_flags = torch.tensor([0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int32).unsqueeze(1)
    trunc_flags = torch.tensor([0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1], dtype=torch.int32).unsqueeze(1)
    advantages = ppo.compute_advantages(rewards, values, end_values, done_flags, trunc_flags)
    assert advantages.size(0) == 20 and advantages.size(1) == 1
    assert advantages.dtype == torch.float32
    assert (advantages - TARGET_20x1).abs().mean() < 1e-3


def test_ragged_golden_bit_exact(golden_dir):
    """Every reference-generated case, including both-flags cells and negative-index wrap."""
    from ppo_amd import ops
    g = np.load(os.path.join(golden_dir, "ref_gae.npz"))
    for i in range(int(g["n_cases"])):
        p = f"c{i}_"
        gamma, lam = g[p + "gl"]
        adv = ops.gae_ragged(*(torch.from_numpy(g[p + k]).cuda() for k in ("r", "v", "ev", "done", "trunc")),
                             float(gamma), float(lam))
        assert adv.is_cuda and np.array_equal(adv.cpu().numpy(), g[p + "adv"]), f"case {i}"


def test_ragged_index_error_reported():
    from ppo_amd import ops
    r = torch.ones(2, 1)
    flags = torch.ones(2, 1, dtype=torch.int32)
    with pytest.raises(IndexError):
        ops.gae_ragged(r, r, torch.tensor([1.0]), flags, flags, 0.99, 0.95)


def rand_case(rng, T, N, p_done=0.02, p_trunc=0.002):
    r = rng.standard_normal((T, N), dtype=np.float32)
    v = rng.standard_normal((T, N), dtype=np.float32)
    d = (rng.random((T, N), dtype=np.float32) < p_done).astype(np.int32)
    tr = ((rng.random((T, N), dtype=np.float32) < p_trunc) & (d == 0)).astype(np.int32)
    tr[T - 1] = 1 - d[T - 1]
    ev = rng.standard_normal((T, N), dtype=np.float32)
    return r, v, ev, d, tr


@pytest.mark.parametrize("T,N", [(1, 1), (7, 3), (8, 64), (128, 64), (125, 4), (256, 33), (33, 1000), (16, 4096), (100, 260), (65, 4100),
                                 (128, 8192)])
def test_dense_vs_oracle_all_variants(T, N):
    from ppo_amd import ops
    rng = np.random.default_rng(T * 1000 + N)
    r, v, ev, d, tr = rand_case(rng, T, N, 0.05, 0.02)
    want, want_ret = cref.gae_dense(r, v, ev, d, tr, 0.99, 0.95, want_returns=True)
    dev = [torch.from_numpy(x).cuda() for x in (r, v, ev, d, tr)]
    got, ret = ops.gae_dense(*dev, 0.99, 0.95, want_returns=True)
    assert np.array_equal(got.cpu().numpy(), want) and np.array_equal(ret.cpu().numpy(), want_ret)
    for variant in range(8):
        vec = 2 if variant in (1, 4, 5, 6) else 1
        if N % vec:
            continue
        got = ops.gae_dense(*dev, 0.99, 0.95, variant=variant)
        assert np.array_equal(got.cpu().numpy(), want), f"variant {variant}"


PACKED_VARIANTS = {0: 1, 1: 2, 2: 1, 3: 1, 4: 1, 5: 2}   # variant -> columns per lane (3..5: time-split kernel)


def _pack_words_np(d, tr):
    T, N = d.shape
    words = np.zeros(((T + 15) // 16, N), dtype=np.uint32)
    for t in range(T):
        words[t // 16] |= (d[t] | (tr[t] << 1)).astype(np.uint32) << np.uint32(2 * (t % 16))
    return words.view(np.int32)


@pytest.mark.parametrize("T,N", [(1, 1), (7, 3), (8, 64), (128, 64), (125, 4), (256, 33), (33, 1000), (16, 4096), (100, 260), (65, 4100),
                                 (128, 8192), (256, 4096), (129, 512), (300, 128), (64, 2)])
def test_packed_flag_forms_vs_oracle_all_variants(T, N):
    """ppo_gae_dense_packed (two flag bits per element; column-per-lane and time-split kernels): bit-exact against
    the oracle's scan of src/ppo.py:452-488 in every instantiation, NaN in the unflagged end values not leaking."""
    from ppo_amd import _lib, ops
    rng = np.random.default_rng(T * 1000 + N + 1)
    r, v, ev, d, tr = rand_case(rng, T, N, 0.05, 0.02)
    want, want_ret = cref.gae_dense(r, v, ev, d, tr, 0.99, 0.95, want_returns=True)
    ev_nan = np.where(tr != 0, ev, np.nan).astype(np.float32)
    words = _pack_words_np(d, tr)
    assert np.array_equal(ops.pack_flag_words(torch.from_numpy(d), torch.from_numpy(tr)).numpy(), words)
    dev = [torch.from_numpy(x).cuda() for x in (r, v, ev_nan, words)]
    got, ret = ops.gae_dense_packed(*dev, 0.99, 0.95, want_returns=True)
    assert np.array_equal(got.cpu().numpy(), want) and np.array_equal(ret.cpu().numpy(), want_ret)
    for variant, vec in PACKED_VARIANTS.items():
        if N % vec:
            with pytest.raises(_lib.PpoAmdError):
                ops.gae_dense_packed(*dev, 0.99, 0.95, variant=variant)
            continue
        got, ret = ops.gae_dense_packed(*dev, 0.99, 0.95, variant=variant, want_returns=True)
        assert np.array_equal(got.cpu().numpy(), want), f"variant {variant}"
        assert np.array_equal(ret.cpu().numpy(), want_ret), f"variant {variant} (returns)"


def test_packed_flag_form_keeps_the_reference_arithmetic_on_special_values():
    """Signed zeros, infinities and NaN go through the scan as in the reference (0 * inf = NaN across an episode
    boundary, -0 preserved): the time-split kernel evaluates gl * (keep * na) as (gl * keep) * na, which must be the
    same f32 in every such case."""
    from ppo_amd import ops
    rng = np.random.default_rng(3)
    T, N = 64, 256
    r, v, ev, d, tr = rand_case(rng, T, N, 0.1, 0.05)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e38, -1e38, 1e-45], dtype=np.float32)
    for x in (r, v, ev):
        mask = rng.random((T, N)) < 0.05
        x[mask] = special[rng.integers(0, len(special), int(mask.sum()))]
    with np.errstate(all="ignore"):
        want = cref.gae_dense(r, v, ev, d, tr, 0.99, 0.95)
    dev = [torch.from_numpy(x).cuda() for x in (r, v, ev, _pack_words_np(d, tr))]
    for variant in PACKED_VARIANTS:
        got = ops.gae_dense_packed(*dev, 0.99, 0.95, variant=variant).cpu().numpy()
        assert np.array_equal(got.view(np.uint32) & 0x7FC00000 == 0x7FC00000, np.isnan(want)), variant   # NaN where the oracle has NaN
        ok = ~np.isnan(want)
        assert np.array_equal(got.view(np.uint32)[ok], want.view(np.uint32)[ok]), variant               # same bits elsewhere (incl. -0)


def test_packed_flag_form_rejects_bad_arguments():
    from ppo_amd import _lib, ops
    lib = _lib.load()
    x = torch.zeros(8, 64, device="cuda")
    f = torch.zeros(1, 64, dtype=torch.int32, device="cuda")
    p = lambda t: t.data_ptr()
    assert lib.ppo_gae_flag_words(8, 64) == 64 and lib.ppo_gae_flag_words(17, 3) == 6 and lib.ppo_gae_flag_words(0, 5) == 0
    assert lib.ppo_gae_dense_packed(p(x), p(x), p(x), None, 0.99, 0.95, p(x), None, 8, 64, None) == 1
    assert lib.ppo_gae_dense_packed(p(x), p(x), p(x), p(f), 0.99, 0.95, p(x), None, -1, 64, None) == 1
    assert lib.ppo_gae_dense_packed(p(x), p(x), p(x), p(f) + 2, 0.99, 0.95, p(x), None, 8, 64, None) == 1   # misaligned words
    assert lib.ppo_gae_dense_packed(p(x), p(x), p(x), p(f), 0.99, 0.95, p(x), None, 0, 64, None) == 0   # empty: nothing to do
    with pytest.raises(AssertionError):
        ops.gae_dense_packed(x, x, x, torch.zeros(8, 64, dtype=torch.int32, device="cuda"), 0.99, 0.95)   # wrong word count


def test_dense_ignores_unflagged_end_values():
    """The dense bootstrap buffer is only defined where trunc is set: NaN elsewhere must not leak."""
    from ppo_amd import ops
    rng = np.random.default_rng(5)
    r, v, ev, d, tr = rand_case(rng, 64, 256, 0.05, 0.02)
    want = cref.gae_dense(r, v, ev, d, tr, 0.99, 0.95)
    ev_nan = np.where(tr != 0, ev, np.nan).astype(np.float32)
    got = ops.gae_dense(*(torch.from_numpy(x).cuda() for x in (r, v, ev_nan, d, tr)), 0.99, 0.95)
    assert np.array_equal(got.cpu().numpy(), want)


def test_dense_equals_ragged_on_rollout_shaped_flags():
    from ppo_amd import ops
    rng = np.random.default_rng(9)
    T, N = 128, 512
    r, v, _, d, tr = rand_case(rng, T, N)
    E = int(((d + tr) > 0).sum())
    ev = rng.standard_normal(E).astype(np.float32)
    dense = cref.ragged_to_dense(ev, d, tr)
    a = ops.gae_ragged(*(torch.from_numpy(x).cuda() for x in (r, v, ev, d, tr)), 0.99, 0.95)
    b = ops.gae_dense(*(torch.from_numpy(x).cuda() for x in (r, v, dense, d, tr)), 0.99, 0.95)
    assert torch.equal(a, b)


def test_full_size_properties():
    """BASELINE config 2 shape (T=128, N=65536): the oracle is too slow to run whole, so check
    (i) a 4096-column slice bit-exact against the oracle, (ii) column independence: any column
    subset computed alone gives the same bits, (iii) the closed form on episode-final rows."""
    from ppo_amd import ops
    T, N = 128, 65536
    rng = np.random.default_rng(1234)
    r, v, ev, d, tr = rand_case(rng, T, N)
    dev = [torch.from_numpy(x).cuda() for x in (r, v, ev, d, tr)]
    adv = ops.gae_dense(*dev, 0.99, 0.95)
    sl = slice(30000, 34096)
    want = cref.gae_dense(r[:, sl], v[:, sl], ev[:, sl], d[:, sl], tr[:, sl], 0.99, 0.95)
    assert np.array_equal(adv[:, sl].cpu().numpy(), want)
    sub = [x[:, 1024:3072].contiguous() for x in dev]
    assert torch.equal(ops.gae_dense(*sub, 0.99, 0.95), adv[:, 1024:3072])
    # where an episode ends: A = r + gamma * trunc * end_value - v   (no dependence on the future)
    rr, vv, ee, dd, tt = dev
    ended = (dd + tt) > 0
    g = torch.tensor(0.99, dtype=torch.float32).cuda()
    closed = rr + g * (tt.float() * torch.where(tt != 0, ee, torch.zeros_like(ee))) - vv
    assert torch.equal(adv[ended], closed[ended])


def test_cpu_tensors_are_staged_not_computed_on_cpu():
    ppo = make_ppo()
    r = torch.ones(4, 1)
    v = torch.tensor([[1.0], [2.0], [3.0], [4.0]])
    d = torch.tensor([[0], [1], [0], [0]], dtype=torch.int32)
    tr = torch.tensor([[0], [1], [0], [1]], dtype=torch.int32)
    adv = ppo.compute_advantages(r, v, torch.tensor([5.0, 6.0]), d, tr)  # both-flags probe, SURVEY.md 8c
    assert adv.device.type == "cpu"
    assert torch.allclose(adv.flatten(), torch.tensor([-1.2778246, -3.4639282, 4.7250700, 2.9400001]), rtol=1e-6)


@pytest.mark.parametrize("kind,N,T", [("cartpole", 200, 37), ("acrobot", 96, 5), ("cartpole", 64, 16), ("cartpole", 4100, 129)])
def test_rollout_kernel_flag_words_at_ragged_lengths(kind, N, T):
    """The flag words the whole-rollout kernel leaves (two bits per element, sixteen steps per word, the last word
    partly filled when T is not a multiple of 16) are the packed i32 flag arrays, and the trainer's GAE on them equals
    ppo_gae_dense on the arrays and the oracle, bit for bit; a second rollout into the same buffers too."""
    from ppo_amd import Agent, SyncVecEnv, ops
    from ppo_amd.envs import DeviceEnv
    torch.manual_seed(1)
    probe = DeviceEnv(kind)
    agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
    ve = SyncVecEnv(lambda: DeviceEnv(kind, seed=3), N, T, 0, 0.0, agent, seed=3)
    for _ in range(2):
        ve.rollout()
        assert ve._ev_ready and ve._flag_words.shape == ((T + 15) // 16, N)
        d, tr = ve.done_flags, ve.trunc_flags
        assert torch.equal(ve._flag_words, ops.pack_flag_words(d, tr))
        want = ops.gae_dense(ve.rewards, ve.values, ve._ev_dense, d, tr, 0.99, 0.95)
        got = ops.gae_dense_packed(ve.rewards, ve.values, ve._ev_dense, ve._flag_words, 0.99, 0.95)
        assert torch.equal(got, want)
        evz = torch.where(tr > 0, ve._ev_dense, torch.zeros_like(ve._ev_dense))
        ref = cref.gae_dense(ve.rewards.cpu().numpy(), ve.values.cpu().numpy(), evz.cpu().numpy(), d.cpu().numpy(),
                             tr.cpu().numpy(), 0.99, 0.95)
        assert np.array_equal(got.cpu().numpy(), ref)
