#!/bin/bash
# A B A B timing for library builds under variants/: CartPole joint (bench regime) and Acrobot with the
# TimeLimit counters spread over [0, 500) (steady state of long episodes) -- development aid
set -e
for rep in 1 2; do
for v in "$@"; do
  cp variants/lib$v.so ppo_amd/libppo_amd.so
  echo "== $v: $(timeout -k 10 120 python3 tools/time_rollout.py cartpole 1 65536 2>&1 | grep -v amdgpu.ids)"
  echo "== $v: spread $(PPO_SPREAD_ELAPSED=1 timeout -k 10 120 python3 tools/time_rollout.py acrobot 1 65536 2>&1 | grep -v amdgpu.ids)"
  echo "== $v: spread $(PPO_SPREAD_ELAPSED=1 timeout -k 10 120 python3 tools/time_rollout.py acrobot 0 65536 2>&1 | grep -v amdgpu.ids)"
done
done
