#!/bin/bash
# GPU box, round 6 batch 2: the fp32 kernel after the K-slice fix; cad_gemm_stream's XCD-aware item order against blockIdx order (noxcd) in the
# layer benchmark at both widths, the order of the arms swapped between rounds; whole-step counters of the default build.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/gemm_f32_bench.py 2>&1 | grep product | tee gpurun_out/gemm_f32_bench.txt
timeout 300 python tools/gemm_f32_bench.py --d-model 128 --T 2048 --reps 20 2>&1 | grep product | tee -a gpurun_out/gemm_f32_bench.txt
LAYER_BENCH_ARGS="" bash tools/ab_layer.sh 3 noxcd default | cut -c1-120
cp gpurun_out/ab_layer.log gpurun_out/ab_xcd_c2.log
LAYER_BENCH_ARGS="--d-model 512 --seqlen 262144 --reps 3" bash tools/ab_layer.sh 2 noxcd default default noxcd | cut -c1-120
cp gpurun_out/ab_layer.log gpurun_out/ab_xcd_c4.log
timeout 900 bash tools/prof_step_pmc.sh > gpurun_out/prof_step_pmc.log 2>&1; grep -n "gemm_stream" gpurun_out/prof_step_pmc/summary.txt | cut -c1-250
cp gpurun_out/prof_step_pmc/summary.txt gpurun_out/step_pmc_summary_default.txt
CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_noxcd.so timeout 900 bash tools/prof_step_pmc.sh > gpurun_out/prof_step_pmc_noxcd.log 2>&1; grep -n "gemm_stream" gpurun_out/prof_step_pmc/summary.txt | cut -c1-250
