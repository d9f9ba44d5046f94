#!/bin/bash
# round-3 GPU session 2: parity of the new paths on the device + A/B against the round-2 kernels + Ph with / without the L-split
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py tests/test_seqpar.py tests/test_downstream_golden.py tests/test_equivariance.py -m gpu -x -q > gpurun_out/pytest_r3b.log 2>&1; tail -3 gpurun_out/pytest_r3b.log
timeout 900 python -m pytest tests/test_configs.py -m gpu -q -s > gpurun_out/pytest_configs_r3b.log 2>&1; grep -n "relative-norm\|passed\|failed\|Error" gpurun_out/pytest_configs_r3b.log | cut -c1-1500 | tail -12
timeout 400 bash tools/ab_layer.sh 2 r2 default
for k in 1 2; do
  CADUCEUS_AMD_LSPLIT=$k timeout 300 python bench.py --model ph --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/bench_ph_lsplit$k.log 2>gpurun_out/bench_ph_lsplit$k.err; tail -1 gpurun_out/bench_ph_lsplit$k.log | cut -c1-2200
done
