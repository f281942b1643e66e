#!/bin/bash
set -u
o=gpurun_out/r03/acrobot
mkdir -p $o
AB_ROUNDS=3 timeout -k 10 400 python3 -u tools/ab_multi.py acrobot 1 65536 variants/libshipped.so variants/libnoband.so variants/libonecos.so > $o/ab_guard_band2.txt 2>&1 || { tail $o/ab_guard_band2.txt; exit 1; }
cat $o/ab_guard_band2.txt
