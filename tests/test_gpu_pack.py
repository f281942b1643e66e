"""GPU: ppo_rollout_pack_params (one launch from the nn.Linear tensors) against the step-by-step
definition of the packed layout in oracle/packref.py, bit for bit, for both network modes, both env
kinds, joint and split networks -- and the split-mode margin it reports."""
import numpy as np
import pytest
import torch

from oracle import packref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("obs,actions,joint", [(4, 2, 1), (4, 2, 0), (6, 3, 1), (6, 3, 0)])
@pytest.mark.parametrize("net_mode", [0, 1, 2], ids=["f32", "f16_split", "bf16x3"])
def test_pack_kernel_equals_the_layout_definition(obs, actions, joint, net_mode):
    from ppo_amd import ops, spaces
    from ppo_amd.ppo import Agent
    torch.manual_seed(obs * 10 + joint)
    agent = Agent(spaces.Box(-np.inf, np.inf, (obs,)), spaces.Discrete(actions), 0, joint, "cuda")
    out = None
    for rnd in range(3):
        with torch.no_grad():
            for p in agent.parameters():
                p.add_(0.3 * torch.randn_like(p))
            if rnd == 2:                                  # tiny and huge values: f16 subnormal / large pieces
                agent.pi_backbone[0].weight.mul_(1e-5)
                agent.pi_backbone[2].weight.mul_(300.0)
        want = packref.pack_dense_agent_reference(agent, net_mode)
        out = ops.pack_dense_agent(agent, net_mode, out=out)     # the second / third pack reuse the buffer
        assert out.dtype == torch.float32 and out.shape == want.shape
        assert out.numel() == ops.rollout_param_floats({4: 0, 6: 1}[obs], joint, net_mode)
        assert torch.equal(out.view(torch.int32), want.view(torch.int32))


def test_split_diag_follows_its_definition():
    from ppo_amd import ops, spaces
    from ppo_amd.ppo import Agent
    torch.manual_seed(1)
    agent = Agent(spaces.Box(-np.inf, np.inf, (4,)), spaces.Discrete(2), 0, 0, "cuda")
    diag = torch.empty(3, device="cuda")

    def want():
        m, g = float("inf"), [0.0, 0.0]
        for bb in (agent.pi_backbone, agent.va_backbone):
            for layer, lin in enumerate((bb[0], bb[2])):
                w = (lin.weight.detach().double() * (2.0 * 1.4426950408889634)).abs()
                small = float((w < 0.125).sum())
                if small:
                    m = min(m, 8.0 * float(w.sum()) / small)
                g[layer] = max(g[layer], float((w * w).sum(1).mean().sqrt()))
        return [m] + g

    for scale in (1.0, 0.1, 1e-3, 3e4):
        with torch.no_grad():
            agent.va_backbone[2].weight.mul_(scale)
        ops.pack_dense_agent(agent, ops.NET_F16_SPLIT, diag_out=diag)
        assert np.allclose(diag.tolist(), want(), rtol=1e-5)
    assert diag.tolist()[2] > ops.SPLIT_ROWNORM2_MAX             # large value-backbone weights: auto must leave the f16 split (for the f32-grade bf16x3 kernels)
    _, mode, ok = ops.resolve_net_mode(agent, ops.NET_AUTO)
    assert mode == ops.NET_AUTO_FALLBACK == ops.NET_BF16X3 and ok is False
    torch.manual_seed(2)
    fresh = Agent(spaces.Box(-np.inf, np.inf, (6,)), spaces.Discrete(3), 0, 1, "cuda")
    _, mode, ok = ops.resolve_net_mode(fresh, ops.NET_AUTO)
    assert mode == ops.NET_F16_SPLIT and ok is True              # a default-initialised network: split kernels
