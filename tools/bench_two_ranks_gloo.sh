#!/bin/bash
# bench.py --gpus 2 with both ranks on ONE GPU over gloo (plumbing rehearsal of the driver's multi-GPU launch line;
# measures nothing about scaling).  Output: gpurun_out/<round>/bench_gpus2.{log,json}
o=gpurun_out/${PPO_ROUND:-r02}
mkdir -p $o
PPO_AMD_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 2 --steps 5 --warmup 2 --settle-steps 5 --num-envs 2048 \
    > $o/bench_gpus2.log 2>&1 || { tail -20 $o/bench_gpus2.log; exit 1; }
grep '^{"metric"' $o/bench_gpus2.log > $o/bench_gpus2.json
python3 - <<PY
import json
d = json.load(open("$o/bench_gpus2.json"))
print("n_gpus", d["n_gpus"], "%.1f M env-steps/s" % (d["value"] / 1e6), d["train_step"]["joint"])
PY
