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
    # a variant is a comma-separated list of: a library name (caduceus_amd/libcaduceus_hip_<name>.so), "default", "env:NAME=VALUE" (a switch)
    unset CADUCEUS_AMD_LIB
    envset=""
    for part in ${v//,/ }; do
      case "$part" in
        default) ;;
        env:*) envset="$envset ${part#env:}" ;;
        *) export CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_$part.so ;;
      esac
    done
    # LAYER_BENCH_ARGS: the layer shape, e.g. "--d-model 512 --seqlen 262144" (configs[4]); default = configs[2]
    env $envset timeout 300 python tools/layer_bench.py --tag "$v" $LAYER_BENCH_ARGS 2>/dev/null | grep layer_ms >> gpurun_out/ab_layer.log
  done
done
unset CADUCEUS_AMD_LIB
cat gpurun_out/ab_layer.log
