#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 bash tools/prof_step.sh
timeout 900 bash tools/prof_scan.sh > gpurun_out/prof_scan.log 2>&1
timeout 900 bash tools/prof_proj.sh
python tools/summarize_prof.py gpurun_out/prof > gpurun_out/prof_summary.txt 2>&1
python tools/summarize_prof.py gpurun_out/prof_proj > gpurun_out/prof_proj_summary.txt 2>&1
python tools/summarize_prof.py gpurun_out/prof_step > gpurun_out/prof_step_summary.txt 2>&1
wc -l gpurun_out/prof_summary.txt gpurun_out/prof_proj_summary.txt gpurun_out/prof_step_summary.txt
timeout 600 python bench.py > gpurun_out/bench_full.log 2>gpurun_out/bench_full.err; tail -1 gpurun_out/bench_full.log | cut -c1-300
