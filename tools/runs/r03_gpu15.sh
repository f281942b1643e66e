#!/bin/bash
# iglp_opt scheduling hints on the bf16x3 forward pass: A/B in fresh processes (correctness of a variant is checked first)
set -u
o=gpurun_out/r03
mkdir -p $o
PPO_AMD_NET=bf16x3 AB_ROUNDS=2 timeout -k 10 500 python3 -u tools/ab_multi.py cartpole 1 65536 variants/libbase.so variants/libiglp0.so variants/libiglp1.so variants/libiglp2.so variants/libiglp3.so > $o/ab_iglp.txt 2>&1 || { tail $o/ab_iglp.txt; exit 1; }
cat $o/ab_iglp.txt
