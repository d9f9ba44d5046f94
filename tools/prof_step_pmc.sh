#!/bin/bash
# PMC passes over ONE training step of a one-layer Caduceus-PS at the configs[2] layer shape (tools/step_families.py --n-layer 1):
# matrix-core busy cycles / MFMA instruction counts of every own kernel in one pass, HBM-side bytes in their own passes
# (no tracing combined with --pmc).  Summary: python tools/summarize_step_pmc.py gpurun_out/prof_step_pmc
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_step_pmc
rm -rf $OUT; mkdir -p $OUT
CMD="python tools/step_families.py --n-layer 1 --reps 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -o step -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -o step -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -o step -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o step -- $CMD > $OUT/trace.log 2>&1
python tools/summarize_step_pmc.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
