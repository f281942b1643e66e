"""Whole-rollout kernel time against the rollout length T at N = 65536 (dispatch timestamps via ppo_prof_arm): the
slope is the step, the intercept the launch's fixed part (parameter load, LDS fill, last bootstrap pass, state
write-back).  usage: time_rollout_T.py [kind]   (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
import torch
from ppo_amd import _lib, Agent, SyncVecEnv
from ppo_amd.envs import DeviceEnv
lib = _lib.load()
kind = sys.argv[1] if len(sys.argv) > 1 else "cartpole"
N = 65536
probe = DeviceEnv(kind)
torch.manual_seed(0)
agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
res = []
for T in (16, 32, 64, 128, 256):
    ve = SyncVecEnv(lambda: DeviceEnv(kind), N, T, 0, 0.0, agent, seed=0)
    ve._device_episode_stats = lambda **kw: None
    for _ in range(max(4, 4096 // T)): ve.rollout()          # settle the clocks
    torch.cuda.synchronize()
    us = []
    for _ in range(12):
        _lib.check(lib.ppo_prof_arm(0, 1), "arm")
        ve.rollout()
        _lib.check(lib.ppo_prof_arm(0, 0), "arm")
        torch.cuda.synchronize()
        out = ctypes.c_float()
        _lib.check(lib.ppo_prof_elapsed_us(0, ctypes.byref(out)), "elapsed")
        us.append(out.value)
    res.append((T, float(np.median(us))))
    print(f"T={T:4d}: {np.median(us):8.1f} us per launch, {np.median(us) / T:6.3f} us per step", flush=True)
Ts = np.array([r[0] for r in res], float); us = np.array([r[1] for r in res])
slope, icpt = np.polyfit(Ts, us, 1)
print(f"fit: {slope:.3f} us per step + {icpt:.1f} us per launch")
