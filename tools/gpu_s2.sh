#!/bin/bash
# Round 4, GPU session 2: the fused conv1d-backward + x_proj-gradient kernel -- parity on the device, in-process A/B on the production
# layer (mixer._FUSED_CONV_XPROJ on / off alternating), per-family profile of the layer, bench.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "fused_conv_xproj or conv1d or model_parity or config2 or layer_trace" > gpurun_out/s2_pytest.log 2>&1; tail -2 gpurun_out/s2_pytest.log
timeout 300 python tools/layer_bench.py --ab _FUSED_CONV_XPROJ --rounds 4 > gpurun_out/s2_ab_fused.log 2>&1; tail -1 gpurun_out/s2_ab_fused.log | cut -c1-400
timeout 200 python tools/layer_bench.py > gpurun_out/s2_layer_on.log 2>&1; tail -1 gpurun_out/s2_layer_on.log | cut -c1-600
CADUCEUS_AMD_FUSED_CONV_XPROJ=0 timeout 200 python tools/layer_bench.py > gpurun_out/s2_layer_off.log 2>&1; tail -1 gpurun_out/s2_layer_off.log | cut -c1-600
timeout 400 python bench.py --cpu-sample 0 > gpurun_out/s2_bench.log 2> gpurun_out/s2_bench.err; tail -1 gpurun_out/s2_bench.log | cut -c1-300
