"""Training at BASELINE configs[1] scale (65 536 CartPole envs x 128 steps) for N updates; one JSON line of metrics per
update goes to the file named by PPO_AMD_METRICS.  usage: train_c2.py [joint] [updates] [epochs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import PPO
from ppo_amd.envs import DeviceEnv
joint = int(sys.argv[1]) if len(sys.argv) > 1 else 1
updates = int(sys.argv[2]) if len(sys.argv) > 2 else 150
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
N, T = 65536, 128
torch.manual_seed(0)
ppo = PPO(0.99, 0.95, 1, 1, 0, joint, True, seed=0)
t0 = time.time()
ppo.train(lambda: DeviceEnv("cartpole"), num_updates=updates, num_envs=N, steps_per_env=T, num_epochs=epochs, batch_size=T * N // 4,
          critic_coef=0.5, entropy_coef=0.01, clip_ratio=0.2, max_grad_norm=0.5, learning_rate=3e-4)
m = ppo.last_metrics
print(f"{updates} updates ({updates * N * T / 1e9:.2f} G env steps) in {time.time() - t0:.1f} s: median return {m['utils/ep_return_0.50']}, "
      f"p95 {m['utils/ep_return_0.95']}, max {m['metrics/ep_return_max']}, fused update {ppo.last_paths['fused_update']}")
