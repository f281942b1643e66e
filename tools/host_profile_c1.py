"""cProfile of PPO.train at BASELINE configs[0] scale (CartPole, 8 envs x 128 steps, 4 epochs of 4 minibatches of 256):
the update loop is launch / host bound there, so this is where its host time goes.  usage: host_profile_c1.py [updates]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import PPO
from ppo_amd.envs import DeviceEnv
updates = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ppo = PPO(0.99, 0.95, 1, 1, 0, 1, True, seed=1)
kw = dict(num_envs=8, steps_per_env=128, num_epochs=4, batch_size=256, critic_coef=0.5, entropy_coef=0.01, clip_ratio=0.2,
          max_grad_norm=0.5, learning_rate=2.5e-4)
ppo.train(lambda: DeviceEnv("cartpole"), num_updates=10, **kw)          # warm-up: first launches, allocator
pr = cProfile.Profile()
pr.enable()
ppo.train(lambda: DeviceEnv("cartpole"), num_updates=updates, **kw)
pr.disable()
print({k: round(v, 1) for k, v in ppo.last_metrics.items() if "per_sec" in k})
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
