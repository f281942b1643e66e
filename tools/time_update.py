"""Fused dense update (ppo_dense_update_grads) vs the PyTorch path of PPO.train_step's inner loop, one minibatch
of B samples: forward + loss + backward (no optimiser).  usage: python tools/time_update.py [B] [kind]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ppo_amd import Agent, ops
from ppo_amd.envs import DeviceEnv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2097152
kind = sys.argv[2] if len(sys.argv) > 2 else "cartpole"
probe = DeviceEnv(kind)
gen = torch.Generator(device="cuda").manual_seed(1)
od, A = probe.observation_space.shape[0], probe.action_space.n
states = torch.randn(B, od, device="cuda", generator=gen)
actions = torch.randint(0, A, (B,), device="cuda", generator=gen, dtype=torch.int32)
old_lp = -torch.rand(B, device="cuda", generator=gen) - 0.3
old_v = torch.randn(B, device="cuda", generator=gen)
adv = torch.randn(B, device="cuda", generator=gen)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


for joint in (1, 0):
    agent = Agent(probe.observation_space, probe.action_space, conv_net=0, joint_net=joint, device="cuda")

    def fused(mode=ops.NET_F32):
        ops.dense_update_grads(agent, states, actions, old_lp, old_v, adv, 0.2, 1, 1, 1.0, 0.5, 0.01, net_mode=mode)

    def torch_path():
        for p in agent.parameters():
            p.grad = None
        logits, nv = agent.policy_and_values(states)
        pl, vl, cf, kl, ent = ops.ppo_losses_discrete(logits, actions, nv, old_lp, old_v, adv, 0.2, 1, 1)
        if joint:
            (pl + 0.5 * vl - 0.01 * ent).backward()
        else:
            (pl - 0.01 * ent).backward(retain_graph=True)
            (0.5 * vl).backward()

    tf, ts, tt = timed(fused), timed(lambda: fused(ops.NET_F16_SPLIT)), timed(torch_path)
    print(f"{kind} {'joint' if joint else 'split'} B={B}: fused {tf:.3f} ms (f32 products)  {ts:.3f} ms (f16-split 64 x 64 products)   "
          f"PyTorch + loss/wgrad kernels {tt:.3f} ms   ({tt / ts:.1f}x)")
