import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import Agent, SyncVecEnv, ops
from ppo_amd.envs import DeviceEnv
kind, N, T, seed = "acrobot", 262144, 128, 31
torch.manual_seed(1)
probe = DeviceEnv(kind)
agent = Agent(probe.observation_space, probe.action_space, conv_net=0, joint_net=1, device="cuda")
with torch.no_grad():
    agent.policy.weight.mul_(50.0); agent.policy.bias.uniform_(-0.3, 0.3); agent.critic.bias.fill_(0.2)
ve = SyncVecEnv(lambda: DeviceEnv(kind, seed=seed), N, T, 0, 0.0, agent, seed=seed)
ve.net_mode = ops.NET_F16_SPLIT
g = torch.Generator(device="cuda").manual_seed(2)
near = torch.randint(0, N, (N // 16,), device="cuda", generator=g)
ve.batch.elapsed[near] = torch.randint(380, 499, (near.numel(),), device="cuda", generator=g, dtype=torch.int32)
for k in range(2):
    ve.rollout()
    trunc = ve.trunc_flags
    trm = trunc > 0
    with torch.no_grad():
        want = agent.get_values(ve._term_obs[trm])
    got = ve._ev_dense[trm]
    err = (got - want).abs()
    bad = err > (2e-6 + 1e-5 * want.abs())
    idx = trm.nonzero()[bad]
    print("rollout", k, "cells", int(trm.sum()), "bad", int(bad.sum()), "max err", float(err.max()), "nan", int(torch.isnan(got).sum()))
    if len(idx):
        print(" t of bad:", sorted(set(idx[:, 0].tolist()))[:20], " waves of bad:", sorted(set((idx[:, 1] // 64).tolist()))[:20], "lanes", sorted(set((idx[:, 1] % 64).tolist()))[:10])
        print(" got", got[bad][:6].tolist(), "want", want[bad][:6].tolist())
