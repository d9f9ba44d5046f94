#!/bin/bash
# Whole-step kernel trace (rocprofv3 --kernel-trace --stats) of bench.py, plus PMC passes restricted to our kernels.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 2 --warmup 1 --cpu-sample 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
CMD2="python tools/scan_bench.py --reps 2 --only-scan"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -o scan -- $CMD2 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc2 -o scan -- $CMD2 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o scan -- $CMD2 > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o scan -- $CMD2 > $OUT/pmc4.log 2>&1
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" | head -30
