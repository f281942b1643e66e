#!/bin/bash
set -u
o=gpurun_out/r03
mkdir -p $o
PPO_AMD_NET=bf16x3 AB_ROUNDS=3 timeout -k 10 500 python3 -u tools/ab_multi.py cartpole 1 65536 variants/libbase.so variants/libtser.so > $o/ab_tser.txt 2>&1 || { tail $o/ab_tser.txt; exit 1; }
cat $o/ab_tser.txt
PPO_AMD_NET=f16_split AB_ROUNDS=2 timeout -k 10 500 python3 -u tools/ab_multi.py cartpole 1 65536 variants/libbase.so > $o/ab_tser_f16.txt 2>&1 || { tail $o/ab_tser_f16.txt; exit 1; }
cat $o/ab_tser_f16.txt
