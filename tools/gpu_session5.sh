#!/bin/bash
# round-3 GPU session 5: occupancy experiments at batch 2 (512 scan workgroups per launch): default (S = 16 / 8, 2 waves per SIMD) vs
# forward S = 8 (128 VGPRs, 40 KB LDS: two workgroups per CU) vs backward S = 4 compiled for 4 / 2 waves per SIMD
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/occ.log
for v in default f8o4 f8o2 s4o4 s4o2 default; do
  if [ "$v" = default ]; then unset CADUCEUS_AMD_LIB; else export CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_$v.so; fi
  timeout 200 python tools/layer_bench.py --batch 2 --reps 4 2>gpurun_out/occ_$v.err | grep layer_ms | sed "s/^/batch2 /" | tee -a gpurun_out/occ.log
  tail -2 gpurun_out/occ_$v.err | cut -c1-300
done
unset CADUCEUS_AMD_LIB
