"""GPU: ppo_dense_update_grads (fused forward + loss + backward + weight gradients of one minibatch,
csrc/update.hip) against PyTorch autograd of the same network and the same loss assembly
(reference src/ppo.py:589-625: get_actions_and_values on the minibatch, ratio :595, compute_losses
:490-527, (policy + c_v value - c_e entropy) * anneal :602-604, backward :606).

Truth is a float64 autograd pass; the bar for the kernel is the distance PyTorch's own f32 autograd pass
keeps from it (summation order over the minibatch), with a floor: per gradient tensor,
|kernel - f64| <= 4 |torch f32 - f64| + 2e-6 max|f64 gradient|.  Loss scalars: 1e-5 relative."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _minibatch(kind, B, gen):
    if kind == "cartpole":
        states = (torch.rand(B, 4, device="cuda", generator=gen) * 2 - 1) * torch.tensor([2.4, 3.0, 0.21, 3.0], device="cuda")
        A = 2
    else:
        th = (torch.rand(B, 2, device="cuda", generator=gen) * 2 - 1) * np.pi
        vel = (torch.rand(B, 2, device="cuda", generator=gen) * 2 - 1) * torch.tensor([4 * np.pi, 9 * np.pi], device="cuda")
        states = torch.stack([th[:, 0].cos(), th[:, 0].sin(), th[:, 1].cos(), th[:, 1].sin(), vel[:, 0], vel[:, 1]], 1).float()
        A = 3
    actions = torch.randint(0, A, (B,), device="cuda", generator=gen, dtype=torch.int32)
    old_lp = -torch.rand(B, device="cuda", generator=gen) * 1.5 - 0.05
    old_v = torch.randn(B, device="cuda", generator=gen)
    adv = torch.randn(B, device="cuda", generator=gen) * 3.0 + 0.5
    return states.contiguous(), actions, old_lp, old_v, adv, A


def _autograd(agent, mb, clip, norm_adv, clip_va, c_pi, c_v, c_ent, dtype):
    """The reference's loss assembly in plain torch ops at `dtype`; returns (scalars, {param name: grad})."""
    from oracle import pyref
    states, actions, old_lp, old_v, adv, _ = mb
    ag = copy.deepcopy(agent).to(dtype)
    for p in ag.parameters():
        p.grad = None
    logits, v = ag.policy_and_values(states.to(dtype))
    dist = torch.distributions.Categorical(logits=logits)
    new_lp = dist.log_prob(actions.long())
    entropy = dist.entropy().mean()
    ratio = torch.exp(new_lp - old_lp.to(dtype))
    pl, vl, cf, kl = pyref.losses_torch(ratio, v, old_v.to(dtype), adv.to(dtype), clip, norm_adv=norm_adv, clip_va_loss=clip_va)
    (c_pi * pl + c_v * vl - c_ent * entropy).backward()
    grads = {n: p.grad.detach().double() for n, p in ag.named_parameters() if p.grad is not None}
    return [float(pl), float(vl), float(cf), float(kl), float(entropy)], grads


MODES = pytest.mark.parametrize("net_mode", [0, 1], ids=["f32", "f16split"])   # PPO_NET_F32, PPO_NET_F16_SPLIT


@MODES
@pytest.mark.parametrize("B", [1, 31, 32, 33, 1000, 262144 + 17])
@pytest.mark.parametrize("joint", [1, 0], ids=["joint", "split"])
@pytest.mark.parametrize("kind", ["cartpole", "acrobot"])
def test_fused_update_matches_autograd(kind, joint, B, net_mode):
    from ppo_amd import Agent, ops
    from ppo_amd.envs import DeviceEnv
    gen = torch.Generator(device="cuda").manual_seed(B * 7 + joint)
    torch.manual_seed(B + 3 * joint)
    probe = DeviceEnv(kind)
    agent = Agent(probe.observation_space, probe.action_space, conv_net=0, joint_net=joint, device="cuda")
    with torch.no_grad():
        agent.policy.weight.mul_(30.0)                 # a policy that is not uniform: ratios on both sides of the clip
        agent.critic.bias.fill_(0.3)
    mb = _minibatch(kind, B, gen)
    clip, c_pi, c_v, c_ent = 0.2, 0.9, 0.45, 0.009
    for norm_adv, clip_va in ((1, 1), (0, 0)):
        if B == 1 and norm_adv:
            continue                                    # std of one sample is NaN in the reference too
        assert ops.dense_update_supported(agent)
        out5, groups = ops.dense_update_grads(agent, *mb[:5], clip, norm_adv, clip_va, c_pi, c_v, c_ent, net_mode=net_mode)
        assert len(groups) == (1 if joint else 2)
        pairs = [pg for grp in groups for pg in grp]
        want64, g64 = _autograd(agent, mb, clip, norm_adv, clip_va, c_pi, c_v, c_ent, torch.float64)
        _, g32 = _autograd(agent, mb, clip, norm_adv, clip_va, c_pi, c_v, c_ent, torch.float32)
        got = out5.cpu().numpy()
        assert np.allclose(got[[0, 1, 4]], np.array(want64)[[0, 1, 4]], rtol=1e-5, atol=2e-7), (got, want64)
        assert abs(got[2] - want64[2]) <= 2.0 / B + 1e-6 and abs(got[3] - want64[3]) <= 1e-5 * abs(want64[3]) + 3e-7
        names = {id(p): n for n, p in agent.named_parameters()}
        seen = set()
        for p, g in pairs:
            n = names[id(p)]
            seen.add(n)
            assert g.shape == p.shape
            ref, t32 = g64[n], g32[n].double()
            err = float((g.double() - ref).abs().max())
            bar = 4.0 * float((t32 - ref).abs().max()) + 2e-6 * float(ref.abs().max()) + 1e-12
            assert err <= bar, (kind, joint, B, norm_adv, net_mode, n, err, bar)
        assert seen == set(g64.keys())                  # every parameter with a gradient got one, and no other


@MODES
def test_fused_update_at_the_bench_minibatch_size(net_mode):
    """B = 2 097 152 (BASELINE configs[1]: T N / 4), joint CartPole network, gathered through idx from the 8 M rows of
    a rollout: same bars as above."""
    from ppo_amd import Agent, ops
    from ppo_amd.envs import DeviceEnv
    gen = torch.Generator(device="cuda").manual_seed(21)
    torch.manual_seed(21)
    probe = DeviceEnv("cartpole")
    agent = Agent(probe.observation_space, probe.action_space, conv_net=0, joint_net=1, device="cuda")
    rows, B = 128 * 65536, 2097152
    full = _minibatch("cartpole", rows, gen)
    idx = torch.randperm(rows, device="cuda", generator=gen)[:B].contiguous()
    out5, groups = ops.dense_update_grads(agent, *full[:5], 0.2, 1, 1, 1.0, 0.5, 0.01, idx=idx, net_mode=net_mode)
    mb = tuple(t[idx].contiguous() for t in full[:5]) + (2,)
    want64, g64 = _autograd(agent, mb, 0.2, 1, 1, 1.0, 0.5, 0.01, torch.float64)
    _, g32 = _autograd(agent, mb, 0.2, 1, 1, 1.0, 0.5, 0.01, torch.float32)
    got = out5.cpu().numpy()
    assert np.allclose(got[[0, 1, 4]], np.array(want64)[[0, 1, 4]], rtol=1e-5, atol=2e-7), (got, want64)
    names = {id(p): n for n, p in agent.named_parameters()}
    for p, g in groups[0]:
        ref, t32 = g64[names[id(p)]], g32[names[id(p)]].double()
        err = float((g.double() - ref).abs().max())
        bar = 4.0 * float((t32 - ref).abs().max()) + 2e-6 * float(ref.abs().max()) + 1e-12
        assert err <= bar, (names[id(p)], err, bar)


@MODES
def test_fused_update_gathers_its_minibatch(net_mode):
    """idx given: the minibatch is rows idx of the flattened rollout tensors (reference src/ppo.py:582-588); same
    result, bit for bit, as the gathered minibatch handed over directly."""
    from ppo_amd import Agent, ops
    from ppo_amd.envs import DeviceEnv
    gen = torch.Generator(device="cuda").manual_seed(9)
    for kind, joint in (("cartpole", 1), ("acrobot", 0)):
        probe = DeviceEnv(kind)
        agent = Agent(probe.observation_space, probe.action_space, conv_net=0, joint_net=joint, device="cuda")
        rows, B = 300000, 70001
        full = _minibatch(kind, rows, gen)
        idx = torch.randperm(rows, device="cuda", generator=gen)[:B].contiguous()
        a5, a = ops.dense_update_grads(agent, *full[:5], 0.2, 1, 1, 1.0, 0.5, 0.01, idx=idx, net_mode=net_mode)
        mb = [t[idx].contiguous() for t in full[:5]]
        b5, b = ops.dense_update_grads(agent, *mb, 0.2, 1, 1, 1.0, 0.5, 0.01, net_mode=net_mode)
        assert torch.equal(a5, b5)
        for ga, gb in zip(a, b):
            for (_, x), (_, y) in zip(ga, gb):
                assert torch.equal(x, y)


@MODES
def test_fused_update_is_deterministic_and_masks_the_tail(net_mode):
    from ppo_amd import Agent, ops
    from ppo_amd.envs import DeviceEnv
    gen = torch.Generator(device="cuda").manual_seed(5)
    probe = DeviceEnv("cartpole")
    agent = Agent(probe.observation_space, probe.action_space, conv_net=0, joint_net=1, device="cuda")
    mb = _minibatch("cartpole", 100003, gen)
    a5, a = ops.dense_update_grads(agent, *mb[:5], 0.2, 1, 1, 1.0, 0.5, 0.01, net_mode=net_mode)
    b5, b = ops.dense_update_grads(agent, *mb[:5], 0.2, 1, 1, 1.0, 0.5, 0.01, net_mode=net_mode)
    assert torch.equal(a5, b5)
    for (_, ga), (_, gb) in zip(a[0], b[0]):
        assert torch.equal(ga, gb)


@pytest.mark.parametrize("joint", [1, 0], ids=["joint", "split"])
def test_trainer_with_fused_update_follows_the_pytorch_path(joint):
    """PPO.train for two updates (CartPole, 4096 envs x 32 steps, 2 epochs of 4 minibatches) with the fused
    update and with the PyTorch MLP around the loss kernel, same seeds: same rollouts (the first update's
    are bit-identical), logged losses equal to 1e-4, parameters equal up to what Adam makes of f32
    summation-order differences in the gradients (a gradient entry at the noise level can flip the sign of
    its first Adam steps: allowed for < 1 % of the entries, bounded by the steps taken)."""
    from ppo_amd import PPO
    from ppo_amd.envs import DeviceEnv
    lr, updates, epochs = 3e-4, 2, 2
    runs = {}
    for fused in (True, False):
        torch.manual_seed(11)
        ppo = PPO(0.99, 0.95, 1, 1, 0, joint, True, seed=11)
        ppo.fused_dense_update = fused
        logged = []
        ppo.minibatch_hook = lambda info, _l=logged: _l.append(
            [float(info[k]) for k in ("policy_loss", "critic_loss", "entropy", "kl_div")])
        ppo.train(lambda: DeviceEnv("cartpole"), num_updates=updates, num_envs=4096, steps_per_env=32, num_epochs=epochs,
                  batch_size=4096 * 32 // 4, critic_coef=0.5, entropy_coef=0.01, clip_ratio=0.2, max_grad_norm=0.5,
                  learning_rate=lr)
        assert ppo.last_paths["fused_update"] == fused
        runs[fused] = (np.array(logged), torch.cat([p.detach().reshape(-1) for p in ppo.agent.parameters()]).cpu().numpy())
    la, lb = runs[True][0], runs[False][0]
    assert la.shape == lb.shape == (updates * epochs * 4, 4)
    assert np.allclose(la[0], lb[0], rtol=1e-5, atol=1e-7)            # first minibatch: same parameters, same samples
    assert np.allclose(la, lb, rtol=2e-3, atol=2e-5), np.abs(la - lb).max()
    pa, pb = runs[True][1], runs[False][1]
    steps = updates * epochs * 4
    d = np.abs(pa - pb)
    assert d.max() <= 2.2 * lr * steps, d.max()
    assert (d > 2e-5).mean() < 0.01, (d > 2e-5).mean()


def test_fused_update_rejects_what_it_does_not_cover():
    """Argument checks of the C entry point (hipErrorInvalidValue = 1, nothing launched), and NaN -- not a silent
    zero log-prob -- for an action outside [0, A), like ppo_loss_logits_fwd_bwd (torch's Categorical raises)."""
    from ppo_amd import Agent, _lib, ops
    from ppo_amd.envs import DeviceEnv
    lib = _lib.load()
    assert lib.ppo_dense_update_grad_floats(5, 2, 3) == 0 and lib.ppo_dense_update_grad_floats(4, 4, 3) == 0
    assert lib.ppo_dense_update_grad_floats(4, 2, 3) == 64 * 4 + 64 + 64 * 64 + 64 + 2 * 64 + 2 + 64 + 1
    probe = DeviceEnv("cartpole")
    agent = Agent(probe.observation_space, probe.action_space, conv_net=0, joint_net=1, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(2)
    mb = _minibatch("cartpole", 4096, gen)
    ws = torch.empty(lib.ppo_dense_update_workspace_bytes(), dtype=torch.uint8, device="cuda")
    g = torch.empty(lib.ppo_dense_update_grad_floats(4, 2, 3), dtype=torch.float32, device="cuda")
    out5 = torch.zeros(5, device="cuda")
    l1, l2 = agent.pi_backbone[0], agent.pi_backbone[2]
    def call(obs=4, A=2, heads=3, B=4096, ws_bytes=None, policy_w=agent.policy.weight.data_ptr(), mode=0):
        return lib.ppo_dense_update_grads(
            obs, A, heads, mode, l1.weight.data_ptr(), l1.bias.data_ptr(), l2.weight.data_ptr(), l2.bias.data_ptr(), policy_w,
            agent.policy.bias.data_ptr(), agent.critic.weight.data_ptr(), agent.critic.bias.data_ptr(), None,
            mb[0].data_ptr(), mb[1].data_ptr(), mb[2].data_ptr(), mb[3].data_ptr(), mb[4].data_ptr(), None, 0.2, 0, 1, B,
            1.0, 0.5, 0.01, out5.data_ptr(), g.data_ptr(), ws.data_ptr(), ws.numel() if ws_bytes is None else ws_bytes, None)
    assert call() == 0 and call(mode=1) == 0
    for bad in (dict(obs=5), dict(A=4), dict(heads=0), dict(heads=4), dict(B=0), dict(ws_bytes=1024), dict(policy_w=None), dict(mode=2)):
        assert call(**bad) == 1, bad
    torch.cuda.synchronize()
    actions = mb[1].clone()
    actions[17] = 2                                                     # CartPole has actions 0 and 1
    out, groups = ops.dense_update_grads(agent, mb[0], actions, mb[2], mb[3], mb[4], 0.2, 0, 1, 1.0, 0.5, 0.01)
    assert torch.isnan(out[0]) and any(torch.isnan(gr).any() for _, gr in groups[0])


@pytest.mark.parametrize("scale", [1e-6, 1.0, 1e6], ids=["tiny", "unit", "huge"])
def test_split_mode_scales_the_loss_derivatives_per_tile(scale):
    """PPO_NET_F16_SPLIT scales dz2 per 32-sample tile by a power of two before it becomes an f16 operand: value
    errors (and so loss derivatives) 1e6 times smaller / larger than usual, and a minibatch that mixes both, leave
    the gradients at the same relative distance from float64 autograd."""
    from ppo_amd import Agent, ops
    from ppo_amd.envs import DeviceEnv
    gen = torch.Generator(device="cuda").manual_seed(77)
    torch.manual_seed(77)
    probe = DeviceEnv("cartpole")
    agent = Agent(probe.observation_space, probe.action_space, conv_net=0, joint_net=1, device="cuda")
    B = 50000
    states, actions, old_lp, old_v, adv, A = _minibatch("cartpole", B, gen)
    old_v = old_v * scale
    adv = adv * scale
    old_v[::7] *= 1e-3                                   # tiles that mix magnitudes
    mb = (states, actions, old_lp, old_v, adv, A)
    out5, groups = ops.dense_update_grads(agent, *mb[:5], 0.2, 0, 0, 1.0, 0.5, 0.01, net_mode=1)
    _, g64 = _autograd(agent, mb, 0.2, 0, 0, 1.0, 0.5, 0.01, torch.float64)
    _, g32 = _autograd(agent, mb, 0.2, 0, 0, 1.0, 0.5, 0.01, torch.float32)
    names = {id(p): n for n, p in agent.named_parameters()}
    for p, g in groups[0]:
        ref, t32 = g64[names[id(p)]], g32[names[id(p)]].double()
        assert bool(torch.isfinite(g).all())
        err = float((g.double() - ref).abs().max())
        bar = 4.0 * float((t32 - ref).abs().max()) + 2e-6 * float(ref.abs().max()) + 1e-12
        assert err <= bar, (names[id(p)], scale, err, bar)
