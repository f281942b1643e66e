"""PPO.train at BASELINE configs[1] scale for a few updates (joint net, one epoch of four 2 M-sample minibatches):
run under rocprofv3 --kernel-trace to see where an update's time goes.  usage: trace_train.py [joint] [updates]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import PPO
from ppo_amd.envs import DeviceEnv
joint = int(sys.argv[1]) if len(sys.argv) > 1 else 1
updates = int(sys.argv[2]) if len(sys.argv) > 2 else 6
N, T = 65536, 128
ppo = PPO(0.99, 0.95, 1, 1, 0, joint, True, seed=0)
import time
walls = []
orig = ppo.train_step
def step(*a, **k):
    walls.append(time.perf_counter())
    return orig(*a, **k)
ppo.train_step = step
ppo.train(lambda: DeviceEnv("cartpole"), num_updates=updates, num_envs=N, steps_per_env=T, num_epochs=1, batch_size=T * N // 4,
          critic_coef=0.5, entropy_coef=0.01, clip_ratio=0.2, max_grad_norm=0.5, learning_rate=3e-4)
if len(walls) > 3:
    d = [b - a for a, b in zip(walls[2:-1], walls[3:])]
    print(f"wall per update (train_step to train_step), median of {len(d)}: {1e3 * sorted(d)[len(d) // 2]:.3f} ms")
print({k: v for k, v in ppo.last_metrics.items() if k.startswith("utils/") and "per_sec" in k})
