"""cProfile of PPO.train at BASELINE configs[1] scale (joint net, one epoch of four 2 M-sample minibatches): the host
time of an update outside its device waits.  usage: host_profile_c2.py [updates]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import PPO
from ppo_amd.envs import DeviceEnv
updates = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N, T = 65536, 128
ppo = PPO(0.99, 0.95, 1, 1, 0, 1, True, seed=1)
kw = dict(num_envs=N, steps_per_env=T, num_epochs=1, batch_size=T * N // 4, critic_coef=0.5, entropy_coef=0.01,
          clip_ratio=0.2, max_grad_norm=0.5, learning_rate=3e-4)
ppo.train(lambda: DeviceEnv("cartpole"), num_updates=10, **kw)
pr = cProfile.Profile()
pr.enable()
ppo.train(lambda: DeviceEnv("cartpole"), num_updates=updates, **kw)
pr.disable()
print({k: round(v, 1) for k, v in ppo.last_metrics.items() if "per_sec" in k})
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
