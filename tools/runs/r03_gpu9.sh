#!/bin/bash
# packed-flag GAE: parity tests, variant timings (cold / cache-resident), PMC traffic of the packed kernel
set -u
export TMPDIR=/tmp
o=gpurun_out/r03/gae8
rm -rf $o; mkdir -p $o
timeout -k 10 500 python -u -m pytest tests/test_gpu_gae.py tests/test_gpu_bench_shapes.py tests/test_abi.py -x -q --timeout 300 --timeout-method thread > $o/tests.log 2>&1
rc=$?; tail -3 $o/tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -u tools/tune_gae_packed.py --sets 4 > $o/tune_cold.txt 2>&1 || { tail $o/tune_cold.txt; exit 1; }
cat $o/tune_cold.txt | grep -v '^{'
timeout -k 10 200 python -u tools/tune_gae_packed.py --sets 1 > $o/tune_hot.txt 2>&1 || { tail $o/tune_hot.txt; exit 1; }
cat $o/tune_hot.txt | grep -v '^{'
timeout -k 10 200 python -u tools/tune_gae_packed.py --sets 4 --trunc-rate 0.002 > $o/tune_cold_trunc.txt 2>&1 || { tail $o/tune_cold_trunc.txt; exit 1; }
cat $o/tune_cold_trunc.txt | grep -v '^{'
