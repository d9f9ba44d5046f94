#!/bin/bash
# Round 4, GPU session 1: parity subset on the new build, the scan_bwd variant A/B (default / unrolled pair loop / arithmetic-only floor)
# on the production layer, and the default bench line.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "scan or equivariance or config2 or bench" > gpurun_out/s1_pytest.log 2>&1; tail -2 gpurun_out/s1_pytest.log
timeout 700 bash tools/ab_layer.sh 3 default u8 floor > /dev/null 2>&1; cp gpurun_out/ab_layer.log gpurun_out/s1_ab_layer.log; cut -c1-200 gpurun_out/s1_ab_layer.log
CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_u8.so timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "scan" > gpurun_out/s1_pytest_u8.log 2>&1; tail -1 gpurun_out/s1_pytest_u8.log
timeout 400 python bench.py > gpurun_out/s1_bench.log 2> gpurun_out/s1_bench.err; tail -1 gpurun_out/s1_bench.log | cut -c1-300
