#!/bin/bash
# GPU box: same-box A/B of the training-step scan launches (forward with saved states + backward) over variant libraries.
# usage: tools/ab_train_scan.sh <rounds> <variant> [<variant> ...]   ("default" = caduceus_amd/libcaduceus_hip.so)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rounds=$1; shift
: > gpurun_out/ab_train_scan.log
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset CADUCEUS_AMD_LIB; else export CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_$v.so; fi
    timeout 120 python tools/train_scan_bench.py 2>/dev/null | grep train_ >> gpurun_out/ab_train_scan.log
  done
done
unset CADUCEUS_AMD_LIB
cat gpurun_out/ab_train_scan.log
