#!/bin/bash
# long C2-scale training runs: which network mode auto picks as the weights grow (utils/net_mode_used per update)
set -u
o=gpurun_out/r03
mkdir -p $o
for joint in 1 0; do
  PPO_AMD_METRICS=$o/c2_training_1000_updates_joint$joint.jsonl timeout -k 10 400 python3 tools/train_c2.py $joint 1000 4 > $o/train_c2_1000_joint$joint.txt 2>&1 || { tail -20 $o/train_c2_1000_joint$joint.txt; exit 1; }
  tail -1 $o/train_c2_1000_joint$joint.txt
  python3 - <<PY
import json
rows = [json.loads(l) for l in open("$o/c2_training_1000_updates_joint$joint.jsonl")]
modes = [r["utils/net_mode_used"] for r in rows]
changes = [(i, modes[i]) for i in range(len(modes)) if i == 0 or modes[i] != modes[i - 1]]
print("updates", len(rows), "mode changes", changes[:12], "counts", {m: modes.count(m) for m in set(modes)})
for i in (0, 99, 299, 599, 999):
    if i < len(rows):
        r = rows[i]; print(i, r["utils/net_mode_used"], "median return", r["utils/ep_return_0.50"], "env-steps/s %.2f G" % (r["utils/env_steps_per_sec"] / 1e9))
PY
done
