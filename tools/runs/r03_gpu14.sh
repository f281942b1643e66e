#!/bin/bash
# statistics on a second stream beside GAE: tests that touch it, then the bench with and without
set -u
o=gpurun_out/r03/overlap
rm -rf $o; mkdir -p $o
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_stats_incremental.py tests/test_gpu_ppo.py tests/test_gpu_bench_shapes.py tests/test_gpu_dp_train.py -x -q --timeout 300 --timeout-method thread > $o/tests.txt 2>&1 || { tail -30 $o/tests.txt; exit 1; }
tail -2 $o/tests.txt
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-train-step > $o/bench_overlap.txt 2>&1 || { tail -20 $o/bench_overlap.txt; exit 1; }
PPO_AMD_STATS_STREAM=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-train-step > $o/bench_serial.txt 2>&1 || { tail -20 $o/bench_serial.txt; exit 1; }
for f in overlap serial; do grep '^{' $o/bench_$f.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['value_f32_grade'], d['sustained']['value'], d['roofline']['avg_launch_us'], {k: v for k, v in d['step_breakdown_us'].items() if k != 'note'})"; done
