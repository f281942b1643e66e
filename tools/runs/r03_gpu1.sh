#!/bin/bash
# round 3, first GPU call: probes for the bf16x3 split, the GPU test suite, the precision table, a bench line
set -o pipefail
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -I ppo_amd/csrc -o /tmp/bf16x3_probe tools/micro/bf16x3_probe.hip && \
timeout -k 10 60 /tmp/bf16x3_probe > gpurun_out/r03/bf16x3_probe.txt 2>&1; echo "probe rc=$?" >> gpurun_out/r03/bf16x3_probe.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/micro/valu_rates.hip && \
timeout -k 10 120 /tmp/valu_rates > gpurun_out/r03/valu_rates.txt 2>&1
cat gpurun_out/r03/bf16x3_probe.txt; tail -3 gpurun_out/r03/valu_rates.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 240 --timeout-method thread > gpurun_out/r03/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r03/pytest_gpu.txt
timeout -k 10 300 python -u tools/net_precision_table.py > gpurun_out/r03/net_precision.txt 2>&1; echo "table rc=$?"
timeout -k 10 300 python -u bench.py > gpurun_out/r03/bench1.json 2> gpurun_out/r03/bench1.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r03/bench1.json
