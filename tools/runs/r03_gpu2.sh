#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
cd "$GRAFT_REPO_ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -I ppo_amd/csrc -o /tmp/bf16x3_probe tools/micro/bf16x3_probe.hip && \
timeout -k 10 60 /tmp/bf16x3_probe > gpurun_out/r03/bf16x3_probe.txt 2>&1; echo "probe rc=$?" >> gpurun_out/r03/bf16x3_probe.txt
cat gpurun_out/r03/bf16x3_probe.txt
