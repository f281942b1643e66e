#!/bin/bash
# bench.py --kernel-trace per library variant (development aid): prints the k_sel_* lines
export TMPDIR=/tmp
for v in "$@"; do
  cp variants/lib$v.so ppo_amd/libppo_amd.so
  d=gpurun_out/r01/tv_$v
  rm -rf $d; mkdir -p $d
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $d.log 2>&1 || { tail -20 $d.log; exit 1; }
  echo "== $v $(grep -o '"ms_per_step": [0-9.]*' $d.log | head -1)"
  python3 tools/prof_summary.py $(ls $d/*/*kernel_trace.csv) 12 | grep k_sel | cut -c1-60,100-160
done
