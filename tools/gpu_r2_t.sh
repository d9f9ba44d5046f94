#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed" gpurun_out/pytest_gpu.log | tail -3; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_gpu.log | head
timeout 600 python bench.py > gpurun_out/bench_full.log 2>gpurun_out/bench_full.err; tail -1 gpurun_out/bench_full.log | cut -c1-200
timeout 900 bash tools/prof_step.sh
timeout 600 python bench.py --model ph --seqlen 1024 --batch 128 --cpu-sample 0 > gpurun_out/bench_c2.log 2>gpurun_out/bench_c2.err; tail -1 gpurun_out/bench_c2.log | cut -c1-200
timeout 900 python bench.py --global-batch 8 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bench_gb8.log 2>gpurun_out/bench_gb8.err; tail -1 gpurun_out/bench_gb8.log | cut -c1-200
timeout 300 python tools/proj_bench.py > gpurun_out/proj_bench.log 2>&1; tail -1 gpurun_out/proj_bench.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
