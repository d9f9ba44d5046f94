#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_proj.py -m gpu -q > gpurun_out/pytest_proj.log 2>&1; tail -3 gpurun_out/pytest_proj.log
timeout 300 python tools/proj_bench.py > gpurun_out/proj_bench.log 2>&1; tail -1 gpurun_out/proj_bench.log | cut -c1-1800
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed" gpurun_out/pytest_gpu.log | tail -3; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_gpu.log | head
timeout 600 python bench.py --cpu-sample 0 > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -1 gpurun_out/bench.log | cut -c1-200
