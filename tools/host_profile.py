import time, torch, sys
sys.path.insert(0, '/root/repo')
from ppo_amd import Agent, SyncVecEnv, PPO, ops
from ppo_amd.envs import DeviceEnv
N, T = 65536, 128
probe = DeviceEnv("cartpole")
agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
ve = SyncVecEnv(lambda: DeviceEnv("cartpole", seed=2), N, T, 0, 0.0, agent, seed=2)
import ppo_amd.ppo as P
ppo = P.PPO.__new__(P.PPO)
for _ in range(60):
    ve.rollout()
torch.cuda.synchronize()
rows = []
for it in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ve.rollout()
    t1 = time.perf_counter()
    adv = ops.gae_dense(ve.rewards, ve.values, ve._ev_dense, ve.done_flags, ve.trunc_flags, 0.99, 0.95)
    t2 = time.perf_counter()
    ve._flush_stats()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0))
import statistics
names = ["rollout() host", "gae host", "stats host", "final sync wait", "total"]
for k, n in enumerate(names):
    print(f"{n:18s} median {statistics.median(r[k] for r in rows) * 1e6:8.1f} us")
