#!/bin/bash
# after the packed-flag GAE: whole GPU suite, smoke, PMC traffic of the new kernel, bench line
set -u
export TMPDIR=/tmp
o=gpurun_out/r03/packed
rm -rf $o; mkdir -p $o
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $o/pytest_gpu.txt 2>&1 || { tail -30 $o/pytest_gpu.txt; exit 1; }
tail -2 $o/pytest_gpu.txt
timeout -k 10 200 python3 __graft_entry__.py smoke > $o/smoke.txt 2>&1 || { tail -20 $o/smoke.txt; exit 1; }
tail -1 $o/smoke.txt
bash tools/pmc_gae_packed.sh > $o/pmc.txt 2>&1 || { tail -20 $o/pmc.txt; exit 1; }
grep -E "traffic_over|ratio|hbm_bytes" $o/pmc.txt
timeout -k 10 600 python3 bench.py > $o/bench.txt 2>&1 || { tail -20 $o/bench.txt; exit 1; }
grep '^{"metric"' $o/bench.txt > $o/bench_line.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03/packed/bench_line.json"))
print({k: d[k] for k in ("value", "ms_per_step", "value_f32_grade", "value_with_exact_f32_products")})
print(d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"]["achieved_at_api_20B_per_elem"])
print(d["step_breakdown_us"])
print(d.get("sustained"))
PY
