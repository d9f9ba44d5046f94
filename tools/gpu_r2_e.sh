#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed" gpurun_out/pytest_gpu.log | tail -3; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_gpu.log | head
timeout 600 python bench.py --cpu-sample 0 > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -1 gpurun_out/bench.log | cut -c1-200
timeout 300 python tools/proj_bench.py > gpurun_out/proj_bench.log 2>&1; tail -1 gpurun_out/proj_bench.log | cut -c1-400
rm -rf gpurun_out/prof; timeout 900 bash tools/prof_scan.sh > gpurun_out/prof_scan.log 2>&1
python tools/summarize_prof.py gpurun_out/prof > gpurun_out/prof_summary.txt 2>&1; grep -A12 "scan_bwd_kernel\|scan_fwd_kernel" gpurun_out/prof_summary.txt | grep "kernel\|FETCH\|WRITE\|SQ_WAIT\|SQ_ACTIVE_INST_ANY\|SQ_WAVE_CYCLES\|LDS_BANK" | head -40
