#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed" gpurun_out/pytest_gpu.log | tail -3; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_gpu.log | head
timeout 600 python bench.py --cpu-sample 0 > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -1 gpurun_out/bench.log | cut -c1-200
CADUCEUS_AMD_FUSED_SOFTPLUS=0 timeout 600 python bench.py --cpu-sample 0 > gpurun_out/bench_nofuse.log 2>gpurun_out/bench_nofuse.err; tail -1 gpurun_out/bench_nofuse.log | cut -c1-200
timeout 600 python bench.py --cpu-sample 0 > gpurun_out/bench2.log 2>gpurun_out/bench2.err; tail -1 gpurun_out/bench2.log | cut -c1-200
