#!/bin/bash
# PMC + kernel-trace profile of the scan kernels at the BASELINE configs[2] shape (run on the GPU box from the repo root).
# Counters are collected in their own passes (no --sys-trace etc. alongside --pmc).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
CMD="python tools/layer_bench.py --reps 2"   # one production mixer layer (dt from the dt_proj epilogue, shared gate), C3 shape
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o scan -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -o scan -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc2 -o scan -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o scan -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o scan -- $CMD > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -50
