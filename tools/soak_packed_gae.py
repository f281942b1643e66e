"""Soak: many consecutive whole-rollout launches at the bench shapes; after each one the flag words must equal the packed
i32 flag arrays and the packed-flag GAE (default kernel and every variant) must equal ppo_gae_dense bit for bit.
usage: soak_packed_gae.py [kind] [N] [rollouts]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ppo_amd import Agent, SyncVecEnv, ops
from ppo_amd.envs import DeviceEnv

kind = sys.argv[1] if len(sys.argv) > 1 else "cartpole"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
rollouts = int(sys.argv[3]) if len(sys.argv) > 3 else 200
T = 128
torch.manual_seed(0)
probe = DeviceEnv(kind)
agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
ve = SyncVecEnv(lambda: DeviceEnv(kind, seed=11), N, T, 0, 0.0, agent, seed=11)
ntr = 0
for k in range(rollouts):
    ve.rollout()
    d, tr = ve.done_flags, ve.trunc_flags
    assert torch.equal(ve._flag_words, ops.pack_flag_words(d, tr)), k
    want = ops.gae_dense(ve.rewards, ve.values, ve._ev_dense, d, tr, 0.99, 0.95)
    assert torch.equal(ops.gae_dense_packed(ve.rewards, ve.values, ve._ev_dense, ve._flag_words, 0.99, 0.95), want), k
    if k % 20 == 0:
        for variant in range(6):
            if N % (2 if variant in (1, 5) else 1) == 0:
                got = ops.gae_dense_packed(ve.rewards, ve.values, ve._ev_dense, ve._flag_words, 0.99, 0.95, variant=variant)
                assert torch.equal(got, want), (k, variant)
    ntr += int((tr[:-1] != 0).sum())
print(f"{kind} N={N}: {rollouts} rollouts, flag words and packed GAE equal the i32 form every time; "
      f"{ntr} mid-rollout truncations seen; median return {ve.median_ep_return}")
