#!/bin/bash
# packed GAE variants at other shapes (narrow batches: does the time-split kernel pay there?)
set -u
o=gpurun_out/r03/gae8
mkdir -p $o
for shape in "256 4096" "128 16384" "512 16384" "128 32768" "2048 1024" "128 262144"; do
  set -- $shape
  echo "== T=$1 N=$2"
  timeout -k 10 120 python -u tools/tune_gae_packed.py --T $1 --N $2 --sets 4 --iters 30 > $o/tune_T$1_N$2.txt 2>&1 || { tail $o/tune_T$1_N$2.txt; exit 1; }
  grep -v '^{' $o/tune_T$1_N$2.txt | grep -v amdgpu.ids
done
