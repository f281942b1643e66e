#!/bin/bash
# bench.py under rocprofv3 --kernel-trace, top kernels by total time (development aid)
export TMPDIR=/tmp
d=gpurun_out/${PPO_ROUND:-r02}/quick
rm -rf $d; mkdir -p $d
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 50 --warmup 100 --no-cpu-baseline --no-train-step > $d.log 2>&1 || { tail -20 $d.log; exit 1; }
grep '^{"metric"' $d.log | cut -c1-200
python3 tools/prof_summary.py $(ls $d/*/*kernel_trace.csv) ${1:-12} | cut -c1-60,100-200
