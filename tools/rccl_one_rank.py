"""RCCL smoke on ONE GPU: a world_size-1 process group with backend "nccl" (= RCCL), the collectives ppo_amd/dist.py
issues (dtypes, shapes, reduce ops, device tensors), and three updates of PPO.train with the data-parallel branches
forced on (world 1, so every collective is an identity).  Catches what a gloo rehearsal cannot: an unsupported dtype /
op in this RCCL build, a host tensor handed to RCCL.  It measures nothing about scaling.  usage: rccl_one_rank.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch
import torch.distributed as td
torch.cuda.set_device(0)
td.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
flat = torch.randn(4675, device=dev); ref = flat.clone()
td.all_reduce(flat, op=td.ReduceOp.SUM); assert torch.equal(flat, ref)
rows = torch.randn(1, 4, 3, dtype=torch.float64, device=dev); ref = rows.clone()
td.all_reduce(rows, op=td.ReduceOp.SUM); assert torch.equal(rows, ref)
y = torch.tensor([1.25], dtype=torch.float64, device=dev)
td.all_reduce(y, op=td.ReduceOp.MAX); assert float(y) == 1.25
k = torch.tensor(0.5, device=dev)
td.all_reduce(k); assert float(k) == 0.5
td.broadcast(flat, 0)
td.barrier()
torch.cuda.synchronize()
print("collectives ok (f32 [4675], f64 [1,4,3], f64 MAX, 0-d f32, broadcast, barrier)")

from ppo_amd import dist as D
from ppo_amd import PPO
from ppo_amd.envs import DeviceEnv
D.is_dist = lambda: True            # world 1, but every data-parallel branch of the trainer runs, over RCCL
runs = {}
for forced in (True, False):
    if not forced:
        D.is_dist = lambda: False
    torch.manual_seed(3)
    ppo = PPO(0.99, 0.95, 1, 1, 0, 1, True, seed=3)
    ppo.train(lambda: DeviceEnv("cartpole"), num_updates=3, num_envs=4096, steps_per_env=32, num_epochs=2, batch_size=4096 * 32 // 4,
              critic_coef=0.5, entropy_coef=0.01, clip_ratio=0.2, max_grad_norm=0.5, learning_rate=3e-4, target_div=0.05)
    runs[forced] = torch.cat([p.detach().reshape(-1) for p in ppo.agent.parameters()]).cpu()
    print("data-parallel branches" if forced else "single-process branches", {k: round(v, 4) for k, v in ppo.last_metrics.items() if k.startswith("losses/")})
assert torch.equal(runs[True], runs[False]), "world-1 collectives must be identities"
print("PPO.train through the RCCL branches == single-process run, bit for bit")
td.destroy_process_group()
