#!/bin/bash
set -u
o=gpurun_out/r03
mkdir -p $o
libs="variants/libbase.so variants/libmaxilp.so variants/libbias0.so variants/libnounclust.so variants/libpostbi.so variants/libmaxmem.so"
PPO_AMD_NET=bf16x3 AB_ROUNDS=2 timeout -k 10 600 python3 -u tools/ab_multi.py cartpole 1 65536 $libs > $o/ab_sched_bf16.txt 2>&1 || { tail $o/ab_sched_bf16.txt; exit 1; }
cat $o/ab_sched_bf16.txt
PPO_AMD_NET=f16_split AB_ROUNDS=2 timeout -k 10 600 python3 -u tools/ab_multi.py cartpole 1 65536 $libs > $o/ab_sched_f16.txt 2>&1 || { tail $o/ab_sched_f16.txt; exit 1; }
cat $o/ab_sched_f16.txt
