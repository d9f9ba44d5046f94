#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_timing.so timeout 300 python tools/phase_timing.py > gpurun_out/phase_timing.txt 2>&1; grep -v "^{" gpurun_out/phase_timing.txt | tail -32
timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed" gpurun_out/pytest_gpu.log | tail -2
timeout 900 bash tools/ab_scan.sh 2 default nopk > /dev/null 2>&1; grep bwd2 gpurun_out/ab_scan.log
