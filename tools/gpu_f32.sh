#!/bin/bash
# GPU box: cad_gemm_f32 -- parity on the device, stand-alone timing against torch.mm, and the step traces of the fp32 configurations
# (BASELINE configs[0]'s shape; a d_model 256 layer stack at L = 131072 in fp32) with the count of library GEMM kernels.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_proj.py tests/test_model_parity.py tests/test_capi.py tests/test_host_logic.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/gemm_f32_bench.py 2>&1 | tail -5 | tee gpurun_out/gemm_f32_bench.txt
timeout 300 python tools/gemm_f32_bench.py --d-model 128 --T 2048 --reps 20 2>&1 | tail -5 | tee -a gpurun_out/gemm_f32_bench.txt
bash tools/prof_step_any.sh c0_ps_d128_n4_L1024_fp32 --d-model 128 --n-layer 4 --seqlen 1024 --dtype fp32 | head -12 | cut -c1-220
bash tools/prof_step_any.sh ps_d256_n2_L131072_fp32 --d-model 256 --n-layer 2 --seqlen 131072 --dtype fp32 | head -14 | cut -c1-220
