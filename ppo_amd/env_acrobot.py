"""Env module for the CLI (reference src/env_acrobot.py:6-9): Acrobot-v1 with episode
statistics, device-resident."""
from .envs import DeviceEnv


def train_fn():
    return DeviceEnv("acrobot", record_stats=True)
