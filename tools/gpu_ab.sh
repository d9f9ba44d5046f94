#!/bin/bash
# GPU box: same-box A/B of one production mixer layer over variant libraries, + the scan parity subset for every library first.
# usage: tools/gpu_ab.sh <rounds> <variant> [<variant> ...]   ("default" = caduceus_amd/libcaduceus_hip.so)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rounds=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset CADUCEUS_AMD_LIB; else export CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_$v.so; fi
  timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "scan" 2>&1 | tail -1 | sed "s/^/$v: /"
done
unset CADUCEUS_AMD_LIB
bash tools/ab_layer.sh $rounds "$@" | cut -c1-330
