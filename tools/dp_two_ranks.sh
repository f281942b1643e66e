#!/bin/bash
# Two ranks of tests/dp_worker.py on ONE GPU over gloo (rehearsal of the data-parallel trainer at a given size).
# usage: tools/dp_two_ranks.sh <num_envs_global> <T> <batch_global> <updates> <out_prefix> [kind] [joint]
set -u
N=$1; T=$2; B=$3; U=$4; OUT=$5
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 PPO_AMD_DIST_BACKEND=gloo LOCAL_RANK=0
RANK=0 timeout -k 5 120 python -u -X faulthandler tests/dp_worker.py "$OUT" "$N" "$T" "$B" "$U" ${6:-cartpole} ${7:-1} > "$OUT.r0.log" 2>&1 &
P0=$!
RANK=1 timeout -k 5 120 python -u -X faulthandler tests/dp_worker.py "$OUT" "$N" "$T" "$B" "$U" ${6:-cartpole} ${7:-1} > "$OUT.r1.log" 2>&1 &
P1=$!
wait $P0; R0=$?
wait $P1; R1=$?
echo "rank0 rc=$R0 rank1 rc=$R1"
