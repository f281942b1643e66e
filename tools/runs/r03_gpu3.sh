#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
cd "$GRAFT_REPO_ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -I ppo_amd/csrc -o /tmp/bf16x3_probe tools/micro/bf16x3_probe.hip && \
timeout -k 10 60 /tmp/bf16x3_probe > gpurun_out/r03/bf16x3_probe.txt 2>&1; echo "probe rc=$?" >> gpurun_out/r03/bf16x3_probe.txt
tail -4 gpurun_out/r03/bf16x3_probe.txt
timeout -k 10 300 python -u tools/net_precision_table.py > gpurun_out/r03/net_precision.txt 2>&1; echo "table rc=$?"
timeout -k 10 300 python -u -m pytest tests/test_gpu_net_precision.py tests/test_gpu_pack.py -q --timeout 240 --timeout-method thread > gpurun_out/r03/pytest_prec.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03/pytest_prec.txt
timeout -k 10 300 python -u bench.py --no-train-step --no-cpu-baseline > gpurun_out/r03/bench2.json 2> gpurun_out/r03/bench2.err; echo "bench rc=$?"
