#!/bin/bash
# Acrobot: price of the termination guard band (shipped vs the same build without it), then the Acrobot bench line
set -u
o=gpurun_out/r03/acrobot
rm -rf $o; mkdir -p $o
AB_ROUNDS=4 timeout -k 10 400 python3 -u tools/ab_multi.py acrobot 1 65536 variants/libshipped.so variants/libnoband.so > $o/ab_guard_band.txt 2>&1 || { tail $o/ab_guard_band.txt; exit 1; }
cat $o/ab_guard_band.txt
timeout -k 10 300 python3 bench.py --env acrobot --no-cpu-baseline > $o/bench_acrobot.txt 2>&1 || { tail -20 $o/bench_acrobot.txt; exit 1; }
grep '^{"metric"' $o/bench_acrobot.txt > $o/bench_line_acrobot.json
python3 -c "
import json; d=json.load(open('$o/bench_line_acrobot.json')); print(d['value'], d['ms_per_step'], d['value_f32_grade'], d['step_breakdown_us'])"
