#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_lat
rm -rf $OUT; mkdir -p $OUT
CMD2="python tools/scan_bench.py --reps 2 --only-scan"
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc1 -o scan -- $CMD2 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_IFETCH_LEVEL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc2 -o scan -- $CMD2 > $OUT/pmc2.log 2>&1
ls $OUT/*/
