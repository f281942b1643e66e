"""GAE kernel variants INSIDE the pipeline (right behind the whole-rollout kernel that wrote their inputs, as in bench.py),
BASELINE configs[1]: kernel durations from rocprofv3 --kernel-trace are the reference; this script times each variant with
an event pair around the launch (the pair adds the same few us to every variant).  usage: tune_gae_pipeline.py [variants]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ppo_amd import Agent, SyncVecEnv, ops
from ppo_amd.envs import DeviceEnv
variants = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "5,3,4,6,7").split(",")]
N, T = 65536, 128
probe = DeviceEnv("cartpole")
agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
ve = SyncVecEnv(lambda: DeviceEnv("cartpole"), N, T, 0, 0.0, agent, seed=0)
out = torch.empty((T, N), device="cuda")
for _ in range(150):
    ve.rollout()
for rep in range(2):
    for v in variants:
        ts = []
        for _ in range(24):
            ve.rollout()
            ve._flush_stats()
            ve.rollout()                      # the GAE launch sits right behind a rollout kernel, as in the bench
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gae_dense(ve.rewards, ve.values, ve._ev_dense, ve.done_flags, ve.trunc_flags, 0.99, 0.95, out=out, variant=v)
            e1.record()
            ve._flush_stats()
            ts.append((e0, e1))
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in ts[4:])
        print(f"variant {v:2d}: median {np.median(us):6.1f} us  min {us[0]:6.1f} us", flush=True)
