#!/bin/bash
# Round-3 evidence, part 2: PMC of the rollout kernel (f16-split, bf16x3), per-config lines, a C2-scale training run,
# bench.py --gpus 2 through its own launcher (gloo, both ranks on the one GPU: plumbing only).
set -o pipefail
export TMPDIR=/tmp PPO_ROUND=r03
o=gpurun_out/r03
mkdir -p $o
PPO_AMD_NET=f16_split bash tools/pmc_rollout.sh f16_split cartpole 1 65536 > /dev/null || { echo pmc f16 failed; exit 1; }
PPO_AMD_NET=bf16x3 bash tools/pmc_rollout.sh bf16x3 cartpole 1 65536 > /dev/null || { echo pmc bf16x3 failed; exit 1; }
head -3 $o/pmc_f16_split/summary.txt; head -3 $o/pmc_bf16x3/summary.txt
bash tools/run_configs.sh || exit 1
PPO_AMD_METRICS=$o/c2_training_150_updates_joint.jsonl timeout -k 10 300 python3 tools/train_c2.py 1 150 4 > $o/train_c2_joint.txt 2>&1 || { tail -20 $o/train_c2_joint.txt; exit 1; }
tail -1 $o/train_c2_joint.txt
PPO_AMD_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --steps 5 --warmup 2 --settle-steps 5 --num-envs 2048 > $o/bench_gpus2.log 2>&1 || { tail -20 $o/bench_gpus2.log; exit 1; }
grep '^{"metric"' $o/bench_gpus2.log > $o/bench_gpus2.json
python3 -c "
import json; d=json.load(open('$o/bench_gpus2.json')); print('n_gpus', d['n_gpus'], 'ranks_seen', d['ranks_seen'], d['rank_devices'], d['backend'], '%.1f M env-steps/s' % (d['value']/1e6), d['train_step'].get('joint'))"
