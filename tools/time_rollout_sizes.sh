#!/bin/bash
# rollout-only timing at small batch sizes, 32- vs 64-env waves (development aid)
for n in 4096 16384 32768 49152 65536; do
  for w in 32 64; do
    PPO_WAVE_ENVS=$w timeout -k 10 120 python3 tools/time_rollout.py ${1:-cartpole} ${2:-1} $n 2>&1 | grep -v amdgpu.ids | sed "s/^/wave$w /" || exit 1
  done
done
