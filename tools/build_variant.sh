#!/bin/bash
# Build variants/lib<name>.so: the shipped objects with rollout.hip recompiled under extra flags (development A/B aid).
# The device assembly of the variant goes through tools/check_spills.py like every shipped object (findings are
# printed, the library is still built: a faulty variant is sometimes the point).
# usage: tools/build_variant.sh <name> [-DFLAG=..] ...
set -e
name=$1; shift
cd "$(dirname "$0")/../ppo_amd/csrc"
tmp=../../variants/tmp_$name
rm -rf $tmp; mkdir -p $tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
    -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -save-temps=obj -c rollout.hip -o $tmp/rollout.o
python3 ../../tools/check_spills.py $tmp/rollout-hip-amdgcn-amd-amdhsa-gfx950.s || echo "check_spills: FINDINGS in variant $name (see above)"
mv $tmp/rollout.o ../../variants/rollout_$name.o
rm -rf $tmp
objs=$(ls *.o | grep -v '^rollout.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib$name.so $objs ../../variants/rollout_$name.o
echo "built variants/lib$name.so"
