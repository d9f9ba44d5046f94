#!/bin/bash
# rocprofv3 --kernel-trace --stats of the whole training step (bench.py), CSV output under gpurun_out/prof_step
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_step
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-300
