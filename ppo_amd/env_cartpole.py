"""Env module for the CLI (reference src/env_cartpole.py:6-9): CartPole-v1 with episode
statistics, device-resident."""
from .envs import DeviceEnv


def train_fn():
    return DeviceEnv("cartpole", record_stats=True)
