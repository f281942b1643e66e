"""Soak of the whole-rollout kernels for non-finite outputs: T = 1 rollouts of 2^18 envs from the precision test's
states (corners of the observation box included), every network mode x env x joint/split, many launches each.
Written after a development build (k_rollout's FULL form compiled for the three-piece scheme) returned NaN log-probs
in 2-7 of a launch's first 1024 waves, at random (a register hipcc's spill code never restored, DESIGN.md 4.2: what
it holds is what the kernel BEFORE left there, so float64 matrix products run between the launches here, as they did in
the test that caught it).  The shipped kernels must show none.
usage: soak_nan.py [lib.so] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ppo_amd import _lib
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 100
import torch
import test_gpu_net_precision as T
from ppo_amd import Agent, ops
from ppo_amd.envs import DeviceEnv
total_bad = 0
for kind in ("cartpole", "acrobot"):
    gen = torch.Generator(device="cuda").manual_seed(11)
    states = T._states(kind, T.NSTATES, gen)
    probe = DeviceEnv(kind)
    for joint in (1, 0):
        for mode, name in ((ops.NET_F16_SPLIT, "f16_split"), (ops.NET_BF16X3, "bf16x3"), (ops.NET_F32, "f32")):
            bad_waves = 0
            for scale in (1.0, 4.0):
                torch.manual_seed(10001)
                agent = Agent(probe.observation_space, probe.action_space, 0, joint, "cuda")
                with torch.no_grad():
                    for bb in [agent.pi_backbone] + ([] if joint else [agent.va_backbone]):
                        for lin in (bb[0], bb[2]):
                            lin.weight.mul_(scale)
                zero_act = torch.zeros(T.NSTATES, dtype=torch.int32, device="cuda")
                for rep in range(launches // 2):
                    T._f64(agent, states, zero_act)                # float64 kernels' leftovers in the register file
                    v, act, lp, _ = T._kernel_forward(kind, agent, states, mode)
                    bad = ~(torch.isfinite(lp) & torch.isfinite(v))
                    bad_waves += int(bad.view(-1, 64).any(1).sum())
            total_bad += bad_waves
            print(f"{kind:8s} joint={joint} {name:9s}: {bad_waves} waves with a non-finite value or log-prob in {launches} launches x {T.NSTATES // 64} waves", flush=True)
print("TOTAL", total_bad)
