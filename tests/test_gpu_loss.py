"""GPU parity: ppo_adv_stats / ppo_loss_fwd_bwd (through the C ABI) vs the oracle, the
reference's known answers and the reference-generated golden vectors (values and autograd
gradients).  Bar: 1e-5 relative (f32), stated per assert."""
import os

import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def make_ppo(**kw):
    from ppo_amd import PPO
    args = dict(discount_factor=0.99, gae_factor=0.95, norm_adv=1, clip_va_loss=0, conv_net=0, joint_network=1,
                use_gpu=False)
    args.update(kw)
    return PPO(**args)


def test_compute_losses():
    """Mirror of the reference's tests/test_ppo.py:90-122."""
    ppo = make_ppo()
    prob_ratios = torch.tensor([1.21, 0.99, 1.01, 1.05, 0.75], dtype=torch.float32)
    curr_values = torch.tensor([2.0, 1.8, 2.4, 0.8, 2.4], dtype=torch.float32)
    prev_values = torch.tensor([1.8, 1.8, 2.0, 1.0, 2.2], dtype=torch.float32)
    advantages = torch.tensor([1.9, 1.9, 2.3, 0.8, 2.3], dtype=torch.float32)
    policy_loss, value_loss, clip_frac, kl_div = ppo.compute_losses(prob_ratios, curr_values, prev_values,
                                                                    advantages, 0.2)
    for t, want in ((policy_loss, 0.0491), (value_loss, 1.5520), (clip_frac, 0.4000), (kl_div, 0.0117)):
        assert t.dtype == torch.float32
        assert t.size() == torch.Size([])
        assert (t - want).abs().mean() < 1e-3
    # full-precision values obtained by running the reference
    assert abs(policy_loss.item() - 0.0491202) < 1e-6 and abs(value_loss.item() - 1.5519998) < 2e-6
    ppo = make_ppo(clip_va_loss=1)
    _, value_loss, _, _ = ppo.compute_losses(prob_ratios, curr_values, prev_values, advantages, 0.2)
    assert abs(value_loss.item() - 1.6319997) < 2e-6


def _rel(a, b):
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-30))


def test_golden_values_and_gradients(golden_dir):
    from ppo_amd import ops
    L = np.load(os.path.join(golden_dir, "ref_loss.npz"))
    for i in range(int(L["n_cases"])):
        p = f"c{i}_"
        norm_adv, clip_va, clip = L[p + "cfg"]
        t = {k: torch.from_numpy(L[p + k]).cuda() for k in ("new_lp", "old_lp", "new_v", "old_v", "adv", "ratio")}
        ref = L[p + "out4"]
        for ratio_input in (0, 1):
            x = t["ratio"] if ratio_input else t["new_lp"]
            out4, g_x, g_v = ops.loss_fwd_bwd(x, None if ratio_input else t["old_lp"], t["new_v"], t["old_v"],
                                              t["adv"], float(clip), int(norm_adv), int(clip_va), ratio_input)
            out4 = out4.cpu().numpy()
            assert np.allclose(out4[:3], ref[:3], rtol=RTOL, atol=1e-7), f"case {i}"
            # kl = mean(rho - 1 - log rho) cancels in f32 per element; abs floor 2e-7
            assert abs(out4[3] - ref[3]) <= RTOL * abs(ref[3]) + 2e-7, f"case {i}"
            gref = L[p + ("g_ratio" if ratio_input else "g_lp")]
            assert _rel(g_x.cpu().numpy(), gref) < RTOL, f"case {i}"
            assert _rel(g_v.cpu().numpy(), L[p + "g_v"]) < RTOL, f"case {i}"


def test_autograd_function_matches_torch_autograd():
    """Gradients through the differentiable wrapper equal torch autograd of the restated formula."""
    ppo = make_ppo(clip_va_loss=1)
    torch.manual_seed(0)
    B = 3001
    old_lp = -torch.rand(B).cuda()
    new_lp = (old_lp + 0.2 * torch.randn(B).cuda()).requires_grad_(True)
    old_v = torch.randn(B).cuda()
    new_v = (old_v + 0.4 * torch.randn(B).cuda()).requires_grad_(True)
    adv = torch.randn(B).cuda()
    ratio = torch.exp(new_lp - old_lp)
    pl, vl, cf, kl = ppo.compute_losses(ratio, new_v, old_v, adv, 0.2)
    assert not cf.requires_grad and not kl.requires_grad
    (pl + 0.5 * vl).backward()
    g_lp, g_v = new_lp.grad.clone(), new_v.grad.clone()

    nl, nv = new_lp.detach().clone().requires_grad_(True), new_v.detach().clone().requires_grad_(True)
    rho = torch.exp(nl - old_lp)
    ret = adv + old_v
    a = (adv - adv.mean()) / (adv.std() + 1e-8)
    pl2 = -torch.minimum(rho * a, torch.clamp(rho, 0.8, 1.2) * a).mean()
    vl2 = 0.5 * torch.maximum((ret - nv) ** 2, (ret - torch.clamp(nv, old_v - 0.2, old_v + 0.2)) ** 2).mean()
    (pl2 + 0.5 * vl2).backward()
    assert abs(pl.item() - pl2.item()) <= RTOL * abs(pl2.item()) + 1e-7
    assert abs(vl.item() - vl2.item()) <= RTOL * abs(vl2.item())
    assert _rel(g_lp.cpu().numpy(), nl.grad.cpu().numpy()) < RTOL
    assert _rel(g_v.cpu().numpy(), nv.grad.cpu().numpy()) < RTOL


@pytest.mark.parametrize("B", [1, 3, 4, 5, 255, 1024, 4097, 1 << 20])
@pytest.mark.parametrize("norm_adv,clip_va", [(1, 0), (0, 1), (1, 1)])
def test_vs_oracle_sizes(B, norm_adv, clip_va):
    from ppo_amd import ops
    if B == 1 and norm_adv:
        pytest.skip("std of one sample is NaN in torch and here (covered below)")
    rng = np.random.default_rng(B + 7 * norm_adv + clip_va)
    old_lp = -np.abs(rng.standard_normal(B)).astype(np.float32)
    new_lp = (old_lp + 0.1 * rng.standard_normal(B)).astype(np.float32)
    old_v = rng.standard_normal(B).astype(np.float32)
    new_v = (old_v + 0.3 * rng.standard_normal(B)).astype(np.float32)
    adv = (2.0 + 3.0 * rng.standard_normal(B)).astype(np.float32)
    want4, wgx, wgv = cref.loss(new_lp, old_lp, new_v, old_v, adv, 0.2, norm_adv, clip_va, 0)
    t = [torch.from_numpy(x).cuda() for x in (new_lp, old_lp, new_v, old_v, adv)]
    out4, g_x, g_v = ops.loss_fwd_bwd(t[0], t[1], t[2], t[3], t[4], 0.2, norm_adv, clip_va, 0)
    out4 = out4.cpu().numpy()
    assert np.allclose(out4[:3], want4[:3], rtol=RTOL, atol=1e-7)
    assert abs(out4[3] - want4[3]) <= RTOL * abs(want4[3]) + 2e-7
    assert _rel(g_x.cpu().numpy(), wgx) < RTOL and _rel(g_v.cpu().numpy(), wgv) < RTOL
    stats = ops.adv_stats(t[4]).cpu().numpy()
    ws = cref.adv_stats(adv)
    assert stats[0] == B and abs(stats[1] - ws[1]) <= 1e-9 * max(1, abs(ws[1]))
    assert abs(stats[2] - ws[2]) <= 1e-9 * max(ws[2], 1e-30) + 1e-12


def test_single_sample_normalisation_is_nan_like_torch():
    from ppo_amd import ops
    one = torch.ones(1).cuda()
    out4, _, _ = ops.loss_fwd_bwd(one, None, one, one, one, 0.2, 1, 0, 1)
    assert torch.isnan(out4[0]) and torch.isnan(torch.ones(1).std())


def test_unaligned_views_take_scalar_path():
    from ppo_amd import ops
    rng = np.random.default_rng(3)
    B = 1001
    base = [torch.from_numpy(rng.standard_normal(B + 1).astype(np.float32)).cuda() for _ in range(5)]
    base[0] = base[0] * 0.1
    base[1] = base[1] * 0.1
    views = [b[1:] for b in base]  # 4-byte aligned only
    want4, wgx, wgv = cref.loss(*(v.cpu().numpy() for v in views), 0.2, 1, 1, 0)
    out4, g_x, g_v = ops.loss_fwd_bwd(*views, 0.2, 1, 1, 0)
    assert np.allclose(out4.cpu().numpy()[:3], want4[:3], rtol=RTOL, atol=1e-7)
    assert _rel(g_x.cpu().numpy(), wgx) < RTOL and _rel(g_v.cpu().numpy(), wgv) < RTOL


def test_deterministic_and_linear_in_B():
    """Run-to-run bit-identical; and clip_frac/kl are means, so concatenating a batch with itself
    leaves all four outputs unchanged (size-independent property at the full C2 minibatch size)."""
    from ppo_amd import ops
    torch.manual_seed(1)
    B = 2_097_152  # T*N/4 of BASELINE config 2
    old_lp = -torch.rand(B).cuda()
    new_lp = old_lp + 0.05 * torch.randn(B).cuda()
    old_v, adv = torch.randn(B).cuda(), torch.randn(B).cuda()
    new_v = old_v + 0.3 * torch.randn(B).cuda()
    a = ops.loss_fwd_bwd(new_lp, old_lp, new_v, old_v, adv, 0.2, 1, 1, 0)
    b = ops.loss_fwd_bwd(new_lp, old_lp, new_v, old_v, adv, 0.2, 1, 1, 0)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    stats = ops.adv_stats(adv)
    twice = [torch.cat([x, x]) for x in (new_lp, old_lp, new_v, old_v, adv)]
    stats2 = torch.stack([stats[0] * 2, stats[1], stats[2] * 2 * (2 * B - 1) / (2 * (B - 1))])  # same unbiased std
    c = ops.loss_fwd_bwd(*twice, 0.2, 1, 1, 0, stats3=stats2)
    assert torch.allclose(c[0], a[0], rtol=RTOL, atol=1e-8)
    assert torch.allclose(c[1][:B] * 2, a[1], rtol=RTOL, atol=1e-12)


# ---- discrete-head front end: ppo_loss_logits_fwd_bwd ------------------------------------------
def _check_logits(out5, gpl, gent, gv, ref5, rpl, rent, rv, tag):
    out5 = out5.cpu().numpy()
    assert np.allclose(out5[:3], ref5[:3], rtol=2 * RTOL, atol=1e-7), tag
    assert abs(out5[3] - ref5[3]) <= RTOL * abs(ref5[3]) + 2e-7, tag
    assert abs(out5[4] - ref5[4]) <= RTOL * abs(ref5[4]) + 1e-7, tag
    for got, want, key in ((gpl, rpl, "g_pl"), (gent, rent, "g_ent"), (gv, rv, "g_v")):
        got = got.cpu().numpy()
        # softmax / log-softmax cancel in f32: 2e-5 relative to the largest gradient entry
        assert np.abs(got - want).max() <= 2 * RTOL * max(np.abs(want).max(), 1e-30) + 1e-9, f"{tag} {key}"


def test_logits_golden_values_and_gradients(golden_dir):
    """vs the reference's Agent.get_actions_and_values (Categorical) + compute_losses + autograd."""
    from ppo_amd import ops
    L = np.load(os.path.join(golden_dir, "ref_loss_logits.npz"))
    for i in range(int(L["n_cases"])):
        p = f"c{i}_"
        norm_adv, clip_va, clip = L[p + "cfg"]
        t = {k: torch.from_numpy(L[p + k]).cuda() for k in ("logits", "actions", "old_lp", "new_v", "old_v", "adv")}
        out5, gpl, gent, gv = ops.loss_logits_fwd_bwd(t["logits"], t["actions"], t["old_lp"], t["new_v"], t["old_v"],
                                                      t["adv"], float(clip), int(norm_adv), int(clip_va))
        _check_logits(out5, gpl, gent, gv, L[p + "out5"], L[p + "g_pl"], L[p + "g_ent"], L[p + "g_v"], f"case {i}")


@pytest.mark.parametrize("A", [2, 3, 5, 8])
@pytest.mark.parametrize("B", [1, 63, 4097, 1 << 20])
def test_logits_vs_oracle_sizes(A, B):
    from ppo_amd import ops
    rng = np.random.default_rng(A * 1000 + B)
    logits = (2.0 * rng.standard_normal((B, A))).astype(np.float32)
    actions = rng.integers(0, A, B).astype(np.int32)
    old_lp = -np.abs(rng.standard_normal(B)).astype(np.float32)
    old_v = rng.standard_normal(B).astype(np.float32)
    new_v = (old_v + 0.3 * rng.standard_normal(B)).astype(np.float32)
    adv = (1.0 + 2.0 * rng.standard_normal(B)).astype(np.float32)
    norm_adv = int(B > 1)
    w5, wpl, went, wv = cref.loss_logits(logits, actions, old_lp, new_v, old_v, adv, 0.2, norm_adv, 1)
    t = [torch.from_numpy(x).cuda() for x in (logits, actions, old_lp, new_v, old_v, adv)]
    out5, gpl, gent, gv = ops.loss_logits_fwd_bwd(*t, 0.2, norm_adv, 1)
    _check_logits(out5, gpl, gent, gv, w5, wpl, went, wv, f"A={A} B={B}")


def test_logits_front_end_equals_two_stage_path_and_autograd():
    """ops.ppo_losses_discrete == Categorical(log_prob, entropy) + ops.ppo_losses, values and the
    gradients that reach the network parameters (joint net, the way train_step combines them)."""
    from ppo_amd import Agent, ops, spaces
    torch.manual_seed(3)
    agent = Agent(spaces.Box(-1.0, 1.0, (6,)), spaces.Discrete(3), conv_net=0, joint_net=1, device="cuda")
    with torch.no_grad():
        agent.policy.weight.mul_(100.0)
    B = 5000
    states = torch.randn(B, 6, device="cuda")
    actions = torch.randint(0, 3, (B,), device="cuda", dtype=torch.int32)
    old_lp = -torch.rand(B, device="cuda")
    old_v = torch.randn(B, device="cuda")
    adv = torch.randn(B, device="cuda")

    def grads(fused):
        agent.zero_grad()
        if fused:
            logits, v = agent.policy_and_values(states)
            pl, vl, cf, kl, ent = ops.ppo_losses_discrete(logits, actions, v, old_lp, old_v, adv, 0.2, 1, 1)
        else:
            _, lp, v, ent = agent.get_actions_and_values(states, actions)
            pl, vl, cf, kl = ops.ppo_losses(lp, v, old_lp, old_v, adv, 0.2, 1, 1, ratio_input=False)
        (pl + 0.5 * vl - 0.01 * ent).backward()
        return [x.item() for x in (pl, vl, cf, kl, ent)], [p.grad.clone() for p in agent.parameters()]

    (s1, g1), (s2, g2) = grads(True), grads(False)
    for a, b in zip(s1, s2):
        assert abs(a - b) <= 2 * RTOL * abs(b) + 2e-7
    for a, b in zip(g1, g2):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-9


def test_logits_rejects_bad_arguments():
    from ppo_amd import ops, _lib
    B = 8
    z = torch.zeros(B, device="cuda")
    with pytest.raises(_lib.PpoAmdError):   # 9 actions: outside the compiled 2..8 range
        ops.loss_logits_fwd_bwd(torch.zeros(B, 9, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda"),
                                z, z, z, z, 0.2, 0, 0)


def test_logits_out_of_range_action_poisons_the_losses():
    """torch's Categorical.log_prob raises on an action outside [0, A) (reference src/ppo.py:212); the
    kernel must not pass it as log-prob 0: the losses and that sample's gradients become NaN, the
    other samples' gradients stay finite."""
    from ppo_amd import ops
    B, A = 300, 3
    g = torch.Generator(device="cuda").manual_seed(1)
    logits = torch.randn(B, A, device="cuda", generator=g)
    actions = torch.randint(0, A, (B,), device="cuda", dtype=torch.int32, generator=g)
    z = torch.randn(B, device="cuda", generator=g)
    for bad in (A, -1, 77):
        act = actions.clone()
        act[17] = bad
        out5, gpl, gent, gv = ops.loss_logits_fwd_bwd(logits, act, -z.abs(), z, z * 0.5, z + 1.0, 0.2, 1, 1)
        assert bool(torch.isnan(out5[0])) and bool(torch.isnan(out5[3]))
        assert bool(torch.isnan(gpl[17]).all())
        keep = torch.ones(B, dtype=torch.bool, device="cuda")
        keep[17] = False
        assert bool(torch.isfinite(gpl[keep]).all()) and bool(torch.isfinite(gv[keep]).all())


def test_adv_stats_gather_is_bit_identical_to_stats_of_the_gathered_copy():
    """ppo_adv_stats_gather(adv, idx) == ppo_adv_stats(adv[idx]) bit for bit (same threads, groups and order of the
    float64 additions), for sizes around the vector width and a C2-sized minibatch."""
    import torch
    from ppo_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(3)
    rows = 128 * 65536
    adv = torch.randn(rows, device="cuda", generator=gen) * 3 + 0.7
    for B in (1, 2, 3, 4, 5, 1023, 70001, rows // 4):
        idx = torch.randint(0, rows, (B,), device="cuda", generator=gen)
        a, b = ops.adv_stats(adv, idx), ops.adv_stats(adv[idx].contiguous())
        assert torch.equal(a, b), (B, a, b)
