"""ppo_clip_adam_step (csrc/optim.hip): nn.utils.clip_grad_norm_ + optim.Adam.step of the reference's update loop
(src/ppo.py:608-609, :732-733) as one launch, against PyTorch's own clip_grad_norm_ + Adam on the same gradients.

Tolerance: the kernel follows torch's formulas operation for operation in f32, but torch's build may contract a
multiply-add and sums the squares in another order, so parameters are compared to a few f32 roundings of the
step taken (|delta p| <= lr per step), not bit for bit."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _agent(kind, joint):
    from ppo_amd import Agent
    from ppo_amd.envs import DeviceEnv
    probe = DeviceEnv(kind)
    return Agent(probe.observation_space, probe.action_space, conv_net=0, joint_net=joint, device="cuda")


def _group(agent, which, gen, scale):
    """A GradGroup like ops.dense_update_grads builds: (parameter, gradient view) pairs over one flat bucket."""
    from ppo_amd import ops
    bb = agent.pi_backbone if which != "va" else agent.va_backbone
    params = [bb[0].weight, bb[0].bias, bb[2].weight, bb[2].bias]
    if which in ("joint", "pi"):
        params += [agent.policy.weight, agent.policy.bias]
    if which in ("joint", "va"):
        params += [agent.critic.weight, agent.critic.bias]
    n = sum(p.numel() for p in params)
    grp = ops.GradGroup()
    grp.flat = scale * torch.randn(n, device="cuda", generator=gen)
    at = 0
    for p in params:
        grp.append((p, grp.flat[at:at + p.numel()].view_as(p)))
        at += p.numel()
    return grp


@pytest.mark.parametrize("scale", [1e-3, 1.0, 50.0], ids=["unclipped", "clipped", "heavily-clipped"])
@pytest.mark.parametrize("kind,joint", [("cartpole", 1), ("acrobot", 1), ("cartpole", 0)])
def test_clip_adam_step_follows_torch(kind, joint, scale):
    from ppo_amd import ops
    torch.manual_seed(3)
    a = _agent(kind, joint)
    b = copy.deepcopy(a)
    lr, max_norm, steps = 3e-4, 0.5, 12
    opt_a = torch.optim.Adam(a.parameters(), lr=lr, eps=1e-5)
    opt_b = torch.optim.Adam(b.parameters(), lr=lr, eps=1e-5)
    assert ops.clip_adam_supported(opt_a)
    gen = torch.Generator(device="cuda").manual_seed(5)
    names = ["joint"] if joint else ["pi", "va"]
    pb_list = list(b.parameters())
    index = {id(p): k for k, p in enumerate(a.parameters())}
    for step in range(steps):
        for which in names:
            ga = _group(a, which, gen, scale * (1.0 + step))
            gb = [(pb_list[index[id(p)]], g.clone()) for p, g in ga]
            v0 = [p._version for p, _ in ga]
            ops.clip_adam_step(opt_a, ga, max_norm)
            assert all(p._version == v + 1 for (p, _), v in zip(ga, v0))     # prepack() keys the packed parameters on it
            opt_b.zero_grad()
            for p, g in gb:
                p.grad = g
            total = torch.nn.utils.clip_grad_norm_(list(b.parameters()), max_norm)
            opt_b.step()
            # the scaled gradients stay behind in the bucket, as clip_grad_norm_ leaves them in .grad
            want = torch.cat([p.grad.reshape(-1) for p, _ in gb])
            assert torch.allclose(ga.flat, want, rtol=1e-5, atol=1e-30), (float(total), (ga.flat - want).abs().max())
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        d = (pa - pb).abs().max().item()
        assert d <= 2e-7 * steps + 1e-6 * lr * steps, (n, d)
    for pa, pb in zip(a.parameters(), b.parameters()):
        sa, sb = opt_a.state[pa], opt_b.state[pb]
        assert float(sa["step"]) == float(sb["step"]) == steps
        # m = m + w (g - m) cancels: absolute tolerance at a few roundings of the largest entry
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=2e-6 * float(sb["exp_avg"].abs().max()))
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-20)


def test_clip_adam_state_is_adams_own_state():
    """The kernel works on torch.optim.Adam's state tensors: state_dict / load_state_dict round-trip, a plain
    opt.step() continues from it, and a loaded state is picked up (the pointer table is rebuilt)."""
    from ppo_amd import ops
    torch.manual_seed(4)
    a = _agent("cartpole", 1)
    opt = torch.optim.Adam(a.parameters(), lr=1e-3, eps=1e-5)
    gen = torch.Generator(device="cuda").manual_seed(6)
    for _ in range(3):
        ops.clip_adam_step(opt, _group(a, "joint", gen, 1.0), 0.5)
    sd = copy.deepcopy(opt.state_dict())
    assert all(float(s["step"]) == 3 for s in sd["state"].values()) and len(sd["state"]) == 8
    b = copy.deepcopy(a)
    opt_b = torch.optim.Adam(b.parameters(), lr=1e-3, eps=1e-5)
    opt_b.load_state_dict(copy.deepcopy(sd))     # (load_state_dict keeps the given tensors: one copy per optimiser)
    opt.load_state_dict(copy.deepcopy(sd))       # replaces the state tensors of `opt` as well
    g = _group(a, "joint", gen, 1.0)
    index = {id(p): k for k, p in enumerate(a.parameters())}
    pb_list = list(b.parameters())
    gb = [(pb_list[index[id(p)]], ga.clone()) for p, ga in g]
    ops.clip_adam_step(opt, g, 0.5)
    for p, gr in gb:
        p.grad = gr
    torch.nn.utils.clip_grad_norm_(list(b.parameters()), 0.5)
    opt_b.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=0, atol=2e-7)            # a few f32 roundings of an O(0.5) weight
        assert float(opt.state[pa]["step"]) == 4
        assert torch.allclose(opt.state[pa]["exp_avg"], opt_b.state[pb]["exp_avg"], rtol=1e-5,
                              atol=2e-6 * float(opt_b.state[pb]["exp_avg"].abs().max()))


def test_clip_adam_follows_a_parameter_whose_storage_moved():
    """The cached pointer table is rebuilt when a parameter's storage changes (p.data = ..., module.to()): the step
    lands in the new storage, the old one is left alone."""
    from ppo_amd import ops
    torch.manual_seed(8)
    a = _agent("cartpole", 1)
    opt = torch.optim.Adam(a.parameters(), lr=1e-2, eps=1e-5)
    gen = torch.Generator(device="cuda").manual_seed(3)
    ops.clip_adam_step(opt, _group(a, "joint", gen, 1.0), 0.5)
    old = a.critic.weight.data
    a.critic.weight.data = old.clone()
    snapshot = old.clone()
    before = a.critic.weight.detach().clone()
    ops.clip_adam_step(opt, _group(a, "joint", gen, 1.0), 0.5)
    torch.cuda.synchronize()
    assert torch.equal(old, snapshot)                                    # nothing written through the stale pointer
    assert (a.critic.weight.detach() - before).abs().max() > 1e-4        # the live parameter took the step


def test_clip_adam_rejects_bad_arguments_and_other_optimisers():
    import ctypes as C
    from ppo_amd import _lib, ops
    lib = _lib.load()
    p, m, v = (torch.zeros(8, device="cuda") for _ in range(3))
    g = torch.ones(8, device="cuda")
    arr = lambda t: (C.c_void_p * 1)(t.data_ptr())
    numel = (C.c_long * 1)(8)
    call = lambda nt=1, step=1, b1=0.9, g_=g.data_ptr(): lib.ppo_clip_adam_step(
        nt, arr(p), arr(m), arr(v), numel, g_, 0.5, 1e-3, b1, 0.999, 1e-5, step, None)
    assert call() == 0
    for bad in (dict(nt=17), dict(step=0), dict(b1=1.0), dict(g_=None)):
        assert call(**bad) == 1, bad
    assert lib.ppo_clip_adam_step(0, None, None, None, None, None, 0.5, 1e-3, 0.9, 0.999, 1e-5, 1, None) == 0
    torch.cuda.synchronize()
    # one step from zero moments with |g| = 1 everywhere: clipped to 0.5 / sqrt(8); Adam's first step is -lr sign(g)
    assert torch.allclose(p, torch.full_like(p, -1e-3), rtol=1e-4)
    assert torch.allclose(g, torch.full_like(g, 0.5 / 8 ** 0.5), rtol=1e-5)
    w = torch.nn.Parameter(torch.zeros(4, device="cuda"))
    assert not ops.clip_adam_supported(torch.optim.Adam([w], weight_decay=0.1))
    assert not ops.clip_adam_supported(torch.optim.Adam([w], amsgrad=True))
    assert not ops.clip_adam_supported(torch.optim.AdamW([w]))
    assert not ops.clip_adam_supported(torch.optim.SGD([w], lr=0.1))


@pytest.mark.parametrize("joint", [1, 0], ids=["joint", "split"])
def test_trainer_with_fused_optimizer_follows_torch_adam(joint):
    """PPO.train for two updates with ppo_clip_adam_step and with clip_grad_norm_ + Adam.step, same seeds: the first
    update's rollouts are bit-identical, every logged loss agrees to 1e-4 relative, parameters to a few f32
    roundings per step; and the rollout of the second update ran with the UPDATED parameters (the version bump
    reaches SyncVecEnv.prepack)."""
    from ppo_amd import PPO
    from ppo_amd.envs import DeviceEnv
    lr, updates, epochs = 3e-4, 2, 2
    runs = {}
    for fused in (True, False):
        torch.manual_seed(12)
        ppo = PPO(0.99, 0.95, 1, 1, 0, joint, True, seed=12)
        ppo.fused_optimizer = fused
        logged = []
        ppo.minibatch_hook = lambda info, _l=logged: _l.append(
            [float(info[k]) for k in ("policy_loss", "critic_loss", "entropy", "kl_div")])
        ppo.train(lambda: DeviceEnv("cartpole"), num_updates=updates, num_envs=4096, steps_per_env=32, num_epochs=epochs,
                  batch_size=4096 * 32 // 4, critic_coef=0.5, entropy_coef=0.01, clip_ratio=0.2, max_grad_norm=0.5,
                  learning_rate=lr)
        assert ppo.last_paths["fused_update"] and ppo.last_paths["fused_optimizer"] == fused
        runs[fused] = (np.array(logged), torch.cat([p.detach().reshape(-1) for p in ppo.agent.parameters()]).cpu().numpy(),
                       ppo.vec_env.values.clone(), ppo.vec_env.log_probs.clone())
    la, lb = runs[True][0], runs[False][0]
    assert la.shape == lb.shape == (updates * epochs * 4, 4)
    assert np.allclose(la[0], lb[0], rtol=1e-6, atol=1e-8)
    assert np.allclose(la, lb, rtol=1e-4, atol=2e-6), np.abs(la - lb).max()
    d = np.abs(runs[True][1] - runs[False][1])
    assert d.max() <= 1e-5, d.max()
    # second rollout's values: those of the updated network in both runs (a stale pack would differ at 1e-2)
    # (an action that flips on a 1e-7 difference in a logit sends that env down another trajectory: allowed for < 1 %)
    for k in (2, 3):
        assert ((runs[True][k] - runs[False][k]).abs() > 2e-4).float().mean() < 0.01


def test_clip_adam_large_group_takes_the_generic_kernel():
    """More than 16 chunks of 1024 elements: the strided kernel instead of the one-round-trip one; same numbers."""
    import ctypes as C
    from ppo_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(9)
    sizes = [20000, 3, 1025]
    p = [torch.randn(n, device="cuda", generator=gen) for n in sizes]
    q = [torch.nn.Parameter(t.clone()) for t in p]
    m = [torch.zeros_like(t) for t in p]
    v = [torch.zeros_like(t) for t in p]
    opt = torch.optim.Adam(q, lr=1e-3, eps=1e-5)
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    numel = (C.c_long * len(sizes))(*sizes)
    for step in range(1, 4):
        g = torch.randn(sum(sizes), device="cuda", generator=gen)
        at = 0
        for t in q:
            t.grad = g[at:at + t.numel()].clone()
            at += t.numel()
        assert lib.ppo_clip_adam_step(len(sizes), arr(p), arr(m), arr(v), numel, g.data_ptr(), 0.5, 1e-3, 0.9, 0.999, 1e-5,
                                      step, None) == 0
        torch.nn.utils.clip_grad_norm_(q, 0.5)
        opt.step()
        assert torch.allclose(g, torch.cat([t.grad for t in q]), rtol=1e-5, atol=1e-30)
    for a, b in zip(p, q):
        assert torch.allclose(a, b, rtol=0, atol=1e-6)
