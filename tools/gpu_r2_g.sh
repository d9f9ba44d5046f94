#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 bash tools/ab_scan.sh 3 default noswz > /dev/null 2>&1; grep scan_bwd2 gpurun_out/ab_scan.log
timeout 300 python tools/proj_bench.py > gpurun_out/proj_bench.log 2>&1; tail -1 gpurun_out/proj_bench.log | cut -c1-1200
timeout 300 python -m pytest tests/test_kernels.py tests/test_configs.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --cpu-sample 0 > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -1 gpurun_out/bench.log | cut -c1-200
