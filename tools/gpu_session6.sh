#!/bin/bash
# round-3 GPU session 6: same-box A/B of the projection kernels (software-pipelined A fragments, batched staging-tile reads)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/ab_proj.log
for r in 1 2; do
  for v in gemmold default; do
    if [ "$v" = default ]; then unset CADUCEUS_AMD_LIB; else export CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_$v.so; fi
    echo "== $v" >> gpurun_out/ab_proj.log
    timeout 200 python tools/proj_bench.py 2>&1 | grep "^in_proj\|^dy\|^dt_proj\|^du\|^x_proj\|^d(dt" >> gpurun_out/ab_proj.log
  done
done
unset CADUCEUS_AMD_LIB
cat gpurun_out/ab_proj.log | cut -c1-220
timeout 300 python -m pytest tests/test_proj.py tests/test_fp8.py -m gpu -x -q 2>&1 | tail -2
