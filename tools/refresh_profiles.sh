#!/bin/bash
# Round evidence in one GPU call: the GPU test suite, the bench line, the same command under
# rocprofv3 --kernel-trace --stats, and the Acrobot line.  Outputs under gpurun_out/<round>/final/.
set -o pipefail
export TMPDIR=/tmp
o=gpurun_out/${PPO_ROUND:-r02}/final
rm -rf $o/prof
mkdir -p $o
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $o/pytest_gpu.txt 2>&1 || { tail -20 $o/pytest_gpu.txt; exit 1; }
tail -2 $o/pytest_gpu.txt
timeout -k 10 600 python3 bench.py > $o/bench.txt 2>&1 || { tail -20 $o/bench.txt; exit 1; }
grep '^{"metric"' $o/bench.txt > $o/bench_line.json
timeout -k 10 300 python3 bench.py --env acrobot --no-cpu-baseline > $o/bench_acrobot.txt 2>&1 || { tail -20 $o/bench_acrobot.txt; exit 1; }
grep '^{"metric"' $o/bench_acrobot.txt > $o/bench_line_acrobot.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-step > $o/bench_under_rocprof.txt 2>&1 || { tail -20 $o/bench_under_rocprof.txt; exit 1; }
grep '^{"metric"' $o/bench_under_rocprof.txt > $o/bench_line_under_rocprof.json
cat $o/bench_line.json
