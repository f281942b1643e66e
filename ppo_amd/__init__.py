"""MI355X-native PPO rollout / GAE / loss path with the Python surface of BradleyGS1/PPO.

    from ppo_amd import PPO, Agent, SyncVecEnv, ScaledBeta

Kernels live in libppo_amd.so (C ABI: include/ppo_amd.h); there is no CPU fallback.
"""
from .ppo import PPO, Agent, ScaledBeta, SyncVecEnv  # noqa: F401
from . import envs, ops, spaces  # noqa: F401

__all__ = ["PPO", "Agent", "ScaledBeta", "SyncVecEnv", "envs", "ops", "spaces"]
