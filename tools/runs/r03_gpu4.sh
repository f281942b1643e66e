#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export PPO_ROUND=r03
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in "$@"; do
  PPO_AMD_NET=$mode bash tools/pmc_rollout.sh $mode cartpole 1 65536 || exit 1
done
