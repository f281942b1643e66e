#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03
timeout -k 10 900 python -u -m pytest tests/test_gpu_dp_train.py tests/test_gpu_net_precision.py tests/test_gpu_stats_incremental.py tests/test_gpu_ppo.py -q --timeout 300 --timeout-method thread -x > gpurun_out/r03/pytest_dp.txt 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r03/pytest_dp.txt
