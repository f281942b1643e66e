#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
mode=$1; shift
{
PPO_AMD_NET=$mode AB_ROUNDS=${AB_ROUNDS:-2} timeout -k 10 400 python3 tools/ab_multi.py ${AB_KIND:-cartpole} ${AB_JOINT:-1} ${AB_N:-65536} "$@"
} 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03/ab_pipe.txt
