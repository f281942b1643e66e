"""Time pack_dense_agent (runs once per rollout in training, whenever the parameters changed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import Agent, ops
from ppo_amd.envs import DeviceEnv
for kind, joint in (("cartpole", 1), ("cartpole", 0), ("acrobot", 0)):
    probe = DeviceEnv(kind)
    agent = Agent(probe.observation_space, probe.action_space, 0, joint, "cuda")
    for mode in (ops.NET_F16_SPLIT, ops.NET_F32):
        for _ in range(3): ops.pack_dense_agent(agent, mode)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ops.pack_dense_agent(agent, mode)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{kind} joint={joint} mode={mode}: {dt*1e3:.3f} ms per pack")
