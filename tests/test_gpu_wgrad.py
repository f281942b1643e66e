"""GPU parity: ppo_linear_wgrad (through the C ABI) vs torch autograd of nn.functional.linear in
float64 -- the weight / bias gradients of the dense network's layers in the update phase
(reference src/ppo.py:606-625 loss.backward() through the Linear layers of :126-145)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K", [(64, 64), (64, 4), (64, 6), (2, 64), (3, 64), (1, 64), (33, 31), (5, 7)])
@pytest.mark.parametrize("B", [1, 63, 64, 65, 4097, 1 << 17])
def test_wgrad_matches_float64_reference(M, K, B):
    from ppo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M * 1000 + K * 10 + B % 7)
    gy = torch.randn(B, M, device="cuda", generator=g)
    x = torch.randn(B, K, device="cuda", generator=g)
    dW, db = ops.linear_wgrad(gy, x)
    wW = gy.double().t() @ x.double()
    wb = gy.double().sum(0)
    scale = float(np.sqrt(B))                      # sums of B random products
    assert float((dW.double() - wW).abs().max()) <= 2e-6 * scale + 1e-6
    assert float((db.double() - wb).abs().max()) <= 2e-6 * scale + 1e-6


def test_wgrad_is_deterministic_and_handles_unaligned_views():
    from ppo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    B = 100_003
    buf = torch.randn(B * 64 + 1, device="cuda", generator=g)
    gy = buf[1:].view(B, 64)                       # 4-byte aligned only: scalar staging path
    x = torch.randn(B, 6, device="cuda", generator=g)
    a = ops.linear_wgrad(gy, x)
    b = ops.linear_wgrad(gy, x)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.allclose(a[0].double(), gy.double().t() @ x.double(), rtol=1e-5, atol=1e-3)


def test_agent_gradients_equal_plain_linear_layers():
    """The Agent built on ops.linear gives the same parameter gradients as torch's own Linear."""
    from ppo_amd import Agent, spaces
    import torch.nn.functional as F
    torch.manual_seed(0)
    agent = Agent(spaces.Box(-1.0, 1.0, (4,)), spaces.Discrete(2), conv_net=0, joint_net=0, device="cuda")
    B = 20_000
    s = torch.randn(B, 4, device="cuda")
    a = torch.randint(0, 2, (B,), device="cuda", dtype=torch.int32)

    def run(custom):
        agent.zero_grad()
        if custom:
            _, lp, v, ent = agent.get_actions_and_values(s, a)
        else:  # same parameters through torch.nn.functional.linear
            def mlp(seq, z):
                z = torch.tanh(F.linear(z, seq[0].weight, seq[0].bias))
                return torch.tanh(F.linear(z, seq[2].weight, seq[2].bias))
            hp, hv = mlp(agent.pi_backbone, s), mlp(agent.va_backbone, s)
            logits = F.linear(hp, agent.policy.weight, agent.policy.bias)
            v = F.linear(hv, agent.critic.weight, agent.critic.bias).flatten()
            d = torch.distributions.Categorical(logits=logits)
            lp, ent = d.log_prob(a), d.entropy().mean()
        ((lp * torch.linspace(-1, 1, B, device="cuda")).mean() + (v ** 2).mean() - 0.01 * ent).backward()
        return [p.grad.clone() for p in agent.parameters()]

    for ga, gb in zip(run(True), run(False)):
        assert float((ga - gb).abs().max()) <= 1e-5 * float(gb.abs().max()) + 1e-8
