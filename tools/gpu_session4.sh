#!/bin/bash
# round-3 GPU session 4: same-box A/B vs the round-2 scan kernels; occupancy experiments (carry-only pass at 2 vs 4 waves / SIMD;
# S = 4 backward variants compiled for 4 and 2 waves / SIMD at batch 2 = 512 workgroups -- TIMING ONLY, their flush is not valid)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 400 bash tools/ab_layer.sh 2 r2scan default
timeout 200 python tools/carry_bench.py 2>&1 | tail -1
for v in default s4o2 s4o4; do
  if [ "$v" = default ]; then unset CADUCEUS_AMD_LIB; else export CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_$v.so; fi
  timeout 200 python tools/layer_bench.py --batch 2 --reps 4 2>/dev/null | grep layer_ms | sed "s/^/batch2 /"
  timeout 200 python tools/layer_bench.py --batch 1 --reps 4 2>/dev/null | grep layer_ms | sed "s/^/batch1 /"
done
unset CADUCEUS_AMD_LIB
