#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
timeout 600 bash tools/ab_scan.sh 2 dma1 default
timeout 600 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -1 gpurun_out/bench.log | cut -c1-600; tail -3 gpurun_out/bench.err
timeout 900 python bench.py --global-batch 8 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bench_gb8.log 2>gpurun_out/bench_gb8.err; tail -1 gpurun_out/bench_gb8.log | cut -c1-500
