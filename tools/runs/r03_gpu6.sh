#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_coexec tools/micro/mfma_coexec.hip && timeout -k 10 120 /tmp/mfma_coexec | tee gpurun_out/r03/mfma_coexec.txt
timeout -k 10 300 python -u -m pytest tests/test_gpu_env.py -q --timeout 240 --timeout-method thread -s 2>&1 | tail -15 | tee gpurun_out/r03/pytest_env.txt
PPO_AMD_NET=f16_split AB_ROUNDS=2 timeout -k 10 300 python3 tools/ab_multi.py acrobot 1 65536 variants/libbase.so variants/libcur.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/ab_acrobot_terminal.txt
