#!/bin/bash
# GPU box: same-box A/B of one production mixer layer (tools/layer_bench.py) over variant libraries.
# usage: tools/ab_layer.sh <rounds> <variant> [<variant> ...]   ("default" = caduceus_amd/libcaduceus_hip.so)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rounds=$1; shift
export CADUCEUS_AMD_ALLOW_TIMING_BUILD=1   # what-if variants (-DSC_WHATIF) are timing builds: the loader refuses them otherwise
: > gpurun_out/ab_layer.log
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    # a variant is a library (caduceus_amd/libcaduceus_hip_<v>.so), "default", or "env:NAME=VALUE" = the default library under that switch
    unset CADUCEUS_AMD_LIB
    envset=""
    case "$v" in
      default) ;;
      env:*) envset="${v#env:}" ;;
      *) export CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_$v.so ;;
    esac
    env $envset timeout 180 python tools/layer_bench.py --tag "$v" 2>/dev/null | grep layer_ms >> gpurun_out/ab_layer.log
  done
done
unset CADUCEUS_AMD_LIB
cat gpurun_out/ab_layer.log
