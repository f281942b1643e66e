#!/bin/bash
set -u
o=gpurun_out/r03
mkdir -p $o
PPO_AMD_NET=bf16x3 AB_ROUNDS=3 timeout -k 10 500 python3 -u tools/ab_multi.py cartpole 1 65536 variants/libbase.so variants/libldsw.so > $o/ab_ldsw.txt 2>&1 || { tail $o/ab_ldsw.txt; exit 1; }
cat $o/ab_ldsw.txt
