#!/bin/bash
# One measured line per BASELINE.json config on one MI355X; outputs under gpurun_out/<round>/configs/.
set -o pipefail
o=gpurun_out/${PPO_ROUND:-r02}/configs
mkdir -p $o
# C1: plumbing run through the CLI, CartPole, 8 envs x 128 steps, 20 updates
PPO_AMD_METRICS=$o/c1_metrics.jsonl PPO_AMD_SEED=1 timeout -k 10 300 python3 -m ppo_amd.ppo_cli c1 env_cartpole 20 8 128 4 256 0.5 0.01 0.2 0.5 0.00025 0.99 0.95 1 1 0 1 --use_gpu 1 > $o/c1.log 2>&1 || { tail -20 $o/c1.log; exit 1; }
tail -1 $o/c1_metrics.jsonl | cut -c1-400
# C3: Acrobot, 262144 envs on one GPU, and the per-GPU shape of the 8-GPU split (32768)
timeout -k 10 300 python3 bench.py --env acrobot --num-envs 262144 --no-cpu-baseline 2>&1 | grep '^{"metric"' > $o/c3_acrobot_262144.json || exit 1
timeout -k 10 300 python3 bench.py --env acrobot --num-envs 32768 --no-cpu-baseline 2>&1 | grep '^{"metric"' > $o/c3_acrobot_32768.json || exit 1
python3 - <<PY
import json
for n in ("c3_acrobot_262144", "c3_acrobot_32768"):
    d = json.load(open("$o/%s.json" % n))
    print(n, "%.2f G env-steps/s, %.3f ms/step" % (d["value"] / 1e9, d["ms_per_step"]))
PY
