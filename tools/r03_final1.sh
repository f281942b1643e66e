#!/bin/bash
# Round-3 evidence, part 1: GPU suite, smoke, bench lines (default, Acrobot, driver flags), kernel trace of the bench.
set -o pipefail
export TMPDIR=/tmp PPO_ROUND=r03
o=gpurun_out/r03/final
rm -rf $o; mkdir -p $o
bash tools/refresh_profiles.sh > $o/refresh.log 2>&1 || { tail -30 $o/refresh.log; exit 1; }
tail -3 $o/pytest_gpu.txt
timeout -k 10 200 python3 __graft_entry__.py smoke > $o/smoke.txt 2>&1 || { tail -20 $o/smoke.txt; exit 1; }
tail -1 $o/smoke.txt
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_driver_flags.txt 2>&1 || { tail -20 $o/bench_driver_flags.txt; exit 1; }
grep '^{"metric"' $o/bench_driver_flags.txt > $o/bench_line_driver_flags_20_5.json
python3 tools/prof_summary.py $(ls $o/prof/*/*kernel_trace.csv | head -1) 14 > $o/bench_kernel_trace_summary.txt
cat $o/bench_kernel_trace_summary.txt | cut -c1-170
python3 - <<'PY'
import json
for n in ("bench_line", "bench_line_acrobot", "bench_line_driver_flags_20_5", "bench_line_under_rocprof"):
    d = json.load(open("gpurun_out/r03/final/%s.json" % n))
    print(n, "%.2f G" % (d["value"] / 1e9), "%.4f ms" % d["ms_per_step"], "f32-grade %.2f G" % (d["value_f32_grade"] / 1e9),
          "gae %.1f us frac %.3f" % (d["roofline"]["avg_launch_us"], d["roofline"]["frac"]), d.get("sustained") and "%.2f G sustained" % (d["sustained"]["value"] / 1e9))
d = json.load(open("gpurun_out/r03/final/bench_line.json"))
print({k: v for k, v in d["step_breakdown_us"].items() if k != "note"})
print(d.get("train_step"))
print(d.get("cpu_baseline"))
PY
