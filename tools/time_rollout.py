"""PPO_ROLLOUT_ABLATE=<bits> needs a library built with -DPPO_ROLLOUT_ABLATE_BUILD (the switches are
compiled out of the shipped kernel).  Time the whole-rollout kernel alone (events around ve.rollout's kernel via rocprof is better; this
uses wall time over several rollouts with the stats disabled)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import _lib
if os.environ.get('PPO_AB_LIB'): _lib.LIB_PATH = os.environ['PPO_AB_LIB']   # A/B runs against another build of the library
from ppo_amd import Agent, SyncVecEnv, ops
from ppo_amd.envs import DeviceEnv
kind = sys.argv[1] if len(sys.argv) > 1 else "cartpole"
joint = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
T = 128
probe = DeviceEnv(kind)
agent = Agent(probe.observation_space, probe.action_space, 0, joint, "cuda")
ve = SyncVecEnv(lambda: DeviceEnv(kind), N, T, 0, 0.0, agent, seed=0)
ve._device_episode_stats = lambda **kw: None
ve.wave_envs = int(os.environ.get('PPO_WAVE_ENVS', '0'))   # 32 / 64: forced
if os.environ.get('PPO_SPREAD_ELAPSED'):   # TimeLimit counters spread over [0, 500): the steady state of long episodes
    ve.batch.elapsed.copy_(torch.randint(0, 500, (N,), device='cuda', dtype=torch.int32))
for _ in range(2): ve.rollout()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 10
e0.record()
for _ in range(K): ve.rollout()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
print(f"{kind} joint={joint} N={N} ablate={os.environ.get('PPO_ROLLOUT_ABLATE','0')}: {ms:.3f} ms per rollout, {ms*1e3/T:.2f} us/step, {T*N/ms/1e6:.2f} G env-steps/s (rollout only)")
