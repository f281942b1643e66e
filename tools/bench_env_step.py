"""Drive ppo_env_step (the per-step API kernel) at several batch sizes; read dispatch durations from
`rocprofv3 --kernel-trace`.  Algorithmic bytes per env-step: CartPole 108 B (+16 B of
episode statistics), Acrobot 116 B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd.envs import DeviceEnvBatch
for kind, na in (("cartpole", 2), ("acrobot", 3)):
    for N in (65536, 1 << 20, 1 << 22):
        b = DeviceEnvBatch(kind, N, seed=0)
        obs = b.reset()
        nxt, term = torch.empty_like(obs), torch.empty_like(obs)
        rew = torch.empty(N, device="cuda"); done = torch.empty(N, dtype=torch.int32, device="cuda"); trunc = torch.empty_like(done)
        g = torch.Generator(device="cuda").manual_seed(0)
        acts = [torch.randint(0, na, (N,), device="cuda", dtype=torch.int32, generator=g) for _ in range(4)]
        for t in range(24):
            b.step(acts[t % 4], nxt, term, rew, done, trunc)
        torch.cuda.synchronize()
        print(kind, N, "24 steps")
