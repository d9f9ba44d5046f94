#!/bin/bash
# GPU box: same-box A/B of the two scan kernels (two-set launch, C3 layer shape) over variant builds of the library.
# usage: tools/ab_scan.sh <rounds> <variant> [<variant> ...]   ("default" = caduceus_amd/libcaduceus_hip.so)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rounds=$1; shift
: > gpurun_out/ab_scan.log
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset CADUCEUS_AMD_LIB; else export CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_$v.so; fi
    python tools/fwd_only_bench.py 2>/dev/null | grep scan_fwd2 >> gpurun_out/ab_scan.log
    python tools/bwd_only_bench.py 2>/dev/null | grep scan_bwd2 >> gpurun_out/ab_scan.log
  done
done
unset CADUCEUS_AMD_LIB
cat gpurun_out/ab_scan.log
