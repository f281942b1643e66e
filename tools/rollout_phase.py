"""Per-rollout duration of the whole-rollout kernel over consecutive rollouts, with the env batch
re-created (all envs back in the same episode phase) half way: separates clock ramp-up from the
effect of the episode phases.  Prints one duration (us, event pair) per rollout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import Agent, SyncVecEnv
from ppo_amd.envs import DeviceEnv
kind = sys.argv[1] if len(sys.argv) > 1 else "cartpole"
N, T, K = 65536, 128, int(sys.argv[2]) if len(sys.argv) > 2 else 40
probe = DeviceEnv(kind)
torch.manual_seed(0)
agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
for phase in range(2):
    ve = SyncVecEnv(lambda: DeviceEnv(kind), N, T, 0, 0.0, agent, seed=phase)
    ve._device_episode_stats = lambda **kw: None
    ev = []
    for _ in range(K):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ve.rollout(); e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    print(f"phase {phase}:", " ".join(f"{a.elapsed_time(b) * 1e3:.0f}" for a, b in ev), flush=True)
    ends = ((ve.done_flags + ve.trunc_flags) > 0).float().mean().item()
    print(f"  episode ends per env-step in the last rollout: {ends:.4f}")
# third run: the SAME env batch after an idle gap (no reset): a jump back up here is the clock / power
# management of the idle GPU, not the episode phases
import time
torch.cuda.synchronize(); time.sleep(0.5)
ev = []
for _ in range(K):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ve.rollout(); e1.record()
    ev.append((e0, e1))
torch.cuda.synchronize()
print("same envs after 0.5 s idle:", " ".join(f"{a.elapsed_time(b) * 1e3:.0f}" for a, b in ev), flush=True)
