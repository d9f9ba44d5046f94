#!/bin/bash
# Round 4, GPU session 4: fused conv/x_proj backward at 16 waves (default) vs 8 waves (cx8) vs the three-kernel path
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "fused_conv_xproj" > gpurun_out/s4_pytest.log 2>&1; tail -2 gpurun_out/s4_pytest.log
for r in 1 2; do
timeout 200 python tools/layer_bench.py 2>/dev/null | grep layer_ms | cut -c1-400
CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_cx8.so timeout 200 python tools/layer_bench.py 2>/dev/null | grep layer_ms | cut -c1-400
CADUCEUS_AMD_FUSED_CONV_XPROJ=0 timeout 200 python tools/layer_bench.py 2>/dev/null | grep layer_ms | cut -c1-400
done > gpurun_out/s4_layers.log 2>&1
cat gpurun_out/s4_layers.log
