"""Episode statistics at bench scale: full scan vs incremental form, event-timed on the rollout's stream.
  after a rollout  : the call that follows a 128-step rollout (CartPole, random policy: ~6 new episodes per env)
  nothing new      : the same call repeated with unchanged rings (fixed cost: LDS clears, flush, ticket, narrowing)
usage: python tools/time_stats.py [num_envs] [kind]"""
import statistics
import sys

import torch

from ppo_amd import Agent, SyncVecEnv, _lib
from ppo_amd.envs import DeviceEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
kind = sys.argv[2] if len(sys.argv) > 2 else "cartpole"
lib = _lib.load()
probe = DeviceEnv(kind)
agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
ve = SyncVecEnv(lambda: DeviceEnv(kind, seed=2), N, 128, 0, 0.0, agent, seed=2)
b = ve.batch
state = torch.zeros(lib.ppo_episode_stats_state_bytes(N, b.ring), dtype=torch.uint8, device="cuda")
ws = torch.zeros(lib.ppo_episode_stats_workspace_bytes(), dtype=torch.uint8, device="cuda")
s5 = torch.full((5,), float("nan"), dtype=torch.float64, device="cuda")
rings = (b.ret_ring.data_ptr(), b.len_ring.data_ptr(), b.ring_cnt.data_ptr(), b.ring, N)


def incr():
    assert lib.ppo_episode_stats_incremental(*rings, s5.data_ptr(), state.data_ptr(), state.numel(), None) == 0


def full():
    assert lib.ppo_episode_stats(*rings, s5.data_ptr(), ws.data_ptr(), ws.numel(), None) == 0


def timed(fn, reps=1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


ve._stats_pending = False
after_i, after_f = [], []
for it in range(40):
    ve.rollout()
    ve._stats_pending = False          # this script calls the statistics itself
    cnt0 = int(b.ring_cnt.sum())
    t_i = timed(incr)
    t_f = timed(full)
    if it >= 20:
        after_i.append(t_i); after_f.append(t_f)
print(f"{kind} N={N}: finished episodes per env and rollout ~ {cnt0 / N / 40:.2f}")
print(f"after a rollout : incremental {statistics.median(after_i):7.1f} us   full scan {statistics.median(after_f):7.1f} us")
print(f"nothing new     : incremental {timed(incr, 50):7.1f} us   full scan {timed(full, 50):7.1f} us  (back-to-back calls)")
