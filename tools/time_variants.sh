#!/bin/bash
# Time the rollout kernel for each library build under variants/, alternating A B A B on one box
# (development aid; each variant is copied over ppo_amd/libppo_amd.so in the scratch copy on the GPU box).
set -e
out=gpurun_out/r01/variants.txt
mkdir -p gpurun_out/r01
: > $out
for rep in 1 2; do
for v in "$@"; do
  cp variants/lib$v.so ppo_amd/libppo_amd.so
  echo "== $v (pass $rep)" >> $out
  for a in "cartpole 1" "cartpole 0" "acrobot 1" "acrobot 0"; do
    timeout -k 10 120 python3 tools/time_rollout.py $a 65536 2>&1 | grep -v amdgpu.ids >> $out
  done
done
done
cat $out
