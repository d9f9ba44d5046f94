#!/bin/bash
# GPU box: same-box A/B of one production mixer layer over an environment switch of the host path.
# usage: tools/gpu_ab_env.sh <rounds> <ENV_VAR> <value> [<value> ...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rounds=$1; var=$2; shift 2
: > gpurun_out/ab_env.log
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    env $var=$v timeout 180 python tools/layer_bench.py 2>/dev/null | grep layer_ms | sed "s/^/$var=$v /" >> gpurun_out/ab_env.log
  done
done
cat gpurun_out/ab_env.log
