#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_proj.py -m gpu -x -q -k "xTw or own_out_proj" 2>&1 | tail -1
for r in 1 2; do
for v in default xtwr4 xtwqo; do
if [ "$v" = default ]; then unset CADUCEUS_AMD_LIB; else export CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_$v.so; fi
timeout 120 python tools/outproj_bench.py 2>/dev/null | tail -1
done; done > gpurun_out/s7_outproj.log; cat gpurun_out/s7_outproj.log
