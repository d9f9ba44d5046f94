#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py tests/test_configs.py tests/test_full_size.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed" gpurun_out/pytest_gpu.log | tail -2
timeout 900 bash tools/ab_train_scan.sh 3 default oldscan > /dev/null 2>&1; cat gpurun_out/ab_train_scan.log
timeout 600 python bench.py --cpu-sample 0 > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -1 gpurun_out/bench.log | cut -c1-200
