#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 600 bash tools/ab_scan.sh 2 r1 nodma default
CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_timing.so timeout 300 python tools/phase_timing.py > gpurun_out/phase_timing.log 2>&1
cat gpurun_out/phase_timing.log | grep -v "^{"
timeout 600 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -1 gpurun_out/bench.log | cut -c1-400
