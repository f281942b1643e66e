#!/bin/bash
# Time the rollout kernel for each library build under variants/ (development aid; each variant is
# copied over ppo_amd/libppo_amd.so in the scratch copy on the GPU box).
set -e
out=gpurun_out/r01/variants.txt
mkdir -p gpurun_out/r01
: > $out
for v in "$@"; do
  cp variants/lib$v.so ppo_amd/libppo_amd.so
  echo "== $v" >> $out
  timeout -k 10 120 python3 tools/time_rollout.py cartpole 1 65536 >> $out 2>&1
  timeout -k 10 120 python3 tools/time_rollout.py cartpole 0 65536 >> $out 2>&1
  timeout -k 10 120 python3 tools/time_rollout.py acrobot 0 65536 >> $out 2>&1
done
cat $out
