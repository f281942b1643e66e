#!/bin/bash
# Two rocprofv3 --pmc passes (8 SQ counters each) over tools/time_update.py; mean per k_update dispatch.
# usage: pmc_update.sh [kernel name]   (default: the f16-split joint CartPole kernel; "...true, true, false>" is the f32 one)
export TMPDIR=/tmp
d=gpurun_out/${PPO_ROUND:-r02}/pmc_update
rm -rf $d; mkdir -p $d
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $d/p1 -- python3 tools/time_update.py > $d/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $d/p2 -- python3 tools/time_update.py > $d/p2.log 2>&1 || exit 1
for p in p1 p2; do python3 tools/pmc_summary.py $(ls $d/$p/*/*counter_collection.csv) "${1:-k_update<4, 2, true, true, true>}"; done | tee $d/summary.txt
