#!/bin/bash
# PMC passes over the projection kernels (tools/proj_bench.py): matrix-core busy cycles and instruction counts in one pass,
# HBM-side bytes in their own passes (no tracing combined with --pmc).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_proj
rm -rf $OUT; mkdir -p $OUT
CMD="python tools/proj_bench.py"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -o proj -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -o proj -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -o proj -- $CMD > $OUT/pmc3.log 2>&1
