import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, numpy as np
from test_gpu_update import _minibatch, _autograd
from ppo_amd import Agent, ops
from ppo_amd.envs import DeviceEnv
gen = torch.Generator(device="cuda").manual_seed(21)
torch.manual_seed(21)
probe = DeviceEnv("cartpole")
agent = Agent(probe.observation_space, probe.action_space, conv_net=0, joint_net=1, device="cuda")
rows, B = 128 * 65536, 2097152
full = _minibatch("cartpole", rows, gen)
idx = torch.randperm(rows, device="cuda", generator=gen)[:B].contiguous()
mb = tuple(t[idx].contiguous() for t in full[:5]) + (2,)
want64, g64 = _autograd(agent, mb, 0.2, 1, 1, 1.0, 0.5, 0.01, torch.float64)
_, g32 = _autograd(agent, mb, 0.2, 1, 1, 1.0, 0.5, 0.01, torch.float32)
names = {id(p): n for n, p in agent.named_parameters()}
for mode in (0, 1):
    out5, groups = ops.dense_update_grads(agent, *full[:5], 0.2, 1, 1, 1.0, 0.5, 0.01, idx=idx, net_mode=mode)
    for p, g in groups[0]:
        ref, t32 = g64[names[id(p)]], g32[names[id(p)]].double()
        d = (g.double() - ref)
        print(mode, names[id(p)], "err %.2e  mean signed %.2e  torch32 err %.2e  max|ref| %.2e" % (float(d.abs().max()), float(d.mean()), float((t32 - ref).abs().max()), float(ref.abs().max())))
