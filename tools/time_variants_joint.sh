#!/bin/bash
# A B A B timing of the CartPole joint rollout kernel for library builds under variants/ (development aid)
set -e
for rep in 1 2; do
for v in "$@"; do
  cp variants/lib$v.so ppo_amd/libppo_amd.so
  echo "== $v: $(timeout -k 10 120 python3 tools/time_rollout.py cartpole 1 65536 2>&1 | grep -v amdgpu.ids)"
done
done
