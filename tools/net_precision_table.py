"""Error of the whole-rollout kernel's network forward against float64, per backbone weight scale:
forced f16-split, three-piece bf16 (PPO_NET_BF16X3), exact-f32 MFMA and PyTorch's f32 forward, with the pack
kernel's split margin (calibrates ops.split_ok; asserted in tests/test_gpu_net_precision.py).  Max abs error
over 2^18 states for each, and the rms error of the value head for the two f32-grade kernels (the maximum of
2^18 draws is a noisy statistic; the rms is not).
    python tools/net_precision_table.py > profiles/r03_net_precision.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import test_gpu_net_precision as T
from ppo_amd import Agent, ops
from ppo_amd.envs import DeviceEnv

print("kind joint scale head | floor margin, rms row norm W1', W2' | split: v lp | bf16x3: v lp | f32 mfma: v lp | torch f32: v lp   (max abs error over 2^18 states) | rms v: split bf16x3 f32mfma torch")
for kind in ("cartpole", "acrobot"):
    gen = torch.Generator(device="cuda").manual_seed(11)
    states = T._states(kind, T.NSTATES, gen)
    probe = DeviceEnv(kind)
    for joint in (1, 0):
        for scale in (1e-4, 1e-3, 1e-2, 0.1, 0.25, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 30.0):
            for head_gain, bias_amp in ((1.0, 0.0), (60.0, 5.0)):
                torch.manual_seed(int(scale * 1e4) + int(head_gain))
                agent = Agent(probe.observation_space, probe.action_space, 0, joint, "cuda")
                with torch.no_grad():
                    for bb in [agent.pi_backbone] + ([] if joint else [agent.va_backbone]):
                        for lin in (bb[0], bb[2]):
                            lin.weight.mul_(scale)
                            if bias_amp:
                                lin.bias.uniform_(-bias_amp, bias_amp)
                    agent.policy.weight.mul_(head_gain)
                diag = torch.empty(3, device="cuda")
                ops.pack_dense_agent(agent, ops.NET_F16_SPLIT, diag_out=diag)
                dg = diag.tolist()
                out, rms = [], []
                for mode in (ops.NET_F16_SPLIT, ops.NET_BF16X3, ops.NET_F32):
                    v, act, lp, _ = T._kernel_forward(kind, agent, states, mode)
                    v64, lp64 = T._f64(agent, states, act)
                    out += [float((v.double() - v64).abs().max()), float((lp.double() - lp64).abs().max())]
                    rms.append(float((v.double() - v64).pow(2).mean().sqrt()))
                act = torch.zeros(T.NSTATES, dtype=torch.int32, device="cuda")
                v64, lp64 = T._f64(agent, states, act)
                vt, lpt = T._f32_torch(agent, states, act)
                out += [float((vt.double() - v64).abs().max()), float((lpt.double() - lp64).abs().max())]
                rms.append(float((vt.double() - v64).pow(2).mean().sqrt()))
                print(f"{kind:8s} {joint} {scale:7.1e} {head_gain:4.0f} | {dg[0]:8.3g} {dg[1]:6.2f} {dg[2]:6.2f} | "
                      + " | ".join(f"{out[i]:.2e} {out[i+1]:.2e}" for i in (0, 2, 4, 6))
                      + " | " + " ".join(f"{x:.2e}" for x in rms), flush=True)
