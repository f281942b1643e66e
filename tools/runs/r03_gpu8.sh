#!/bin/bash
# config-5 shape tests, then ONE diagnostic world-2 run with gloo handed device tensors (the round-2 hang's setting),
# collective order check and a stack-dumping watchdog on.
set -u
mkdir -p gpurun_out/r03
timeout -k 10 400 python -u -m pytest tests/test_gpu_config5.py -x -v --timeout 300 --timeout-method thread > gpurun_out/r03/config5.log 2>&1
echo "config5 rc=$?"; tail -5 gpurun_out/r03/config5.log
export PPO_AMD_GLOO_DEVICE_TENSORS=1 PPO_AMD_DIST_DEBUG=1 PPO_AMD_DP_WATCHDOG=80
timeout -k 10 150 bash tools/dp_two_ranks.sh 4096 32 32768 6 gpurun_out/r03/dpdev cartpole 0
echo "dpdev rc=$?"
tail -30 gpurun_out/r03/dpdev.r0.log; echo ----; tail -30 gpurun_out/r03/dpdev.r1.log
