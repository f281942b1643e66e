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

# ---- the same bracket inside PPO.train_step (after an update phase), phase by phase
import numpy as np
del ve
torch.cuda.empty_cache()
marks = []
orig_sync = P._sync
ppo = P.PPO(0.99, 0.95, 1, 1, 0, 1, True, seed=0)
orig_train_step = ppo.train_step


def timed_train_step(*a, **k):
    ve = ppo.vec_env
    names, t = [], []
    def wrap(obj, name):
        f = getattr(obj, name)
        def g(*aa, **kk):
            t0 = time.perf_counter(); r = f(*aa, **kk); t.append((name, t0, time.perf_counter())); return r
        setattr(obj, name, g)
        return f
    saved = [(ve, "rollout", wrap(ve, "rollout")), (ppo, "_rollout_advantages", wrap(ppo, "_rollout_advantages")),
             (ve, "_flush_stats", wrap(ve, "_flush_stats")), (ve, "_alloc", wrap(ve, "_alloc")), (ve, "_net", wrap(ve, "_net")),
             (ops, "rollout_fused", wrap(ops, "rollout_fused"))]
    def sync_timed():
        t0 = time.perf_counter(); orig_sync(); t.append(("_sync", t0, time.perf_counter()))
    P._sync = sync_timed
    r = orig_train_step(*a, **k)
    P._sync = orig_sync
    for o, n, f in saved: setattr(o, n, f)
    marks.append(t)
    return r


ppo.train_step = timed_train_step
ppo.train(lambda: DeviceEnv("cartpole"), num_updates=6, num_envs=N, steps_per_env=T, num_epochs=1, batch_size=T * N // 4,
          critic_coef=0.5, entropy_coef=0.01, clip_ratio=0.2, max_grad_norm=0.5, learning_rate=3e-4)
print("inside PPO.train_step (updates 2..6), us from the start of rollout():")
for t in marks[1:]:
    r0 = [x for x in t if x[0] == "rollout"][0][1]
    print("  " + "  ".join(f"{n}:{(a - r0) * 1e6:.0f}-{(b - r0) * 1e6:.0f}" for n, a, b in t if a >= r0 - 1e-3)[:400])
print("  env_steps_per_sec", ppo.last_metrics["utils/env_steps_per_sec"] / 1e9, "G")
