#!/bin/bash
# HBM traffic of the packed-flag GAE kernel by PMC, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes, FETCH_SIZE doubled on gfx950, the doubling calibrated on a kernel of the same access width
# whose read bytes are known (the dense i32 form k_gae<1,8,true>: five [T,N] arrays of 4 B read in full).
# 12 launches over 4 rotating buffer sets (HBM-cold).  Output: gpurun_out/r03/gaepmc/ + a JSON summary.
set -u
export TMPDIR=/tmp
o=gpurun_out/r03/gaepmc
rm -rf $o; mkdir -p $o
args="--done-rate 0.045 --trunc-rate 0"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/fetch -- python3 tools/pmc_gae.py --packed $args > $o/fetch.log 2>&1 || { tail $o/fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/write -- python3 tools/pmc_gae.py --packed $args > $o/write.log 2>&1 || { tail $o/write.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/calib -- python3 tools/pmc_gae.py --variant 0 $args > $o/calib.log 2>&1 || { tail $o/calib.log; exit 1; }
python3 tools/pmc_gae_json.py $o > $o/r03_gae_packed_pmc_traffic.json && cat $o/r03_gae_packed_pmc_traffic.json
