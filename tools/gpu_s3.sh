#!/bin/bash
# Round 4, GPU session 3: fused conv/x_proj backward with pipelined loads (diagnostic of the dx mismatch, A/B), out_proj on cad_proj_xTw
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/dbg_conv_xproj.py > gpurun_out/s3_dbg.log 2>&1; cat gpurun_out/s3_dbg.log | cut -c1-300
timeout 600 python -m pytest tests -m gpu -x -q -k "xTw or own_out_proj" > gpurun_out/s3_pytest.log 2>&1; tail -2 gpurun_out/s3_pytest.log
timeout 300 python tools/layer_bench.py --ab _FUSED_CONV_XPROJ --rounds 4 > gpurun_out/s3_ab_fused.log 2>&1; tail -1 gpurun_out/s3_ab_fused.log | cut -c1-400
timeout 300 python tools/layer_bench.py --ab _OWN_OUT_PROJ --rounds 4 > gpurun_out/s3_ab_outproj.log 2>&1; tail -1 gpurun_out/s3_ab_outproj.log | cut -c1-400
timeout 200 python tools/layer_bench.py > gpurun_out/s3_layer_on.log 2>&1; tail -1 gpurun_out/s3_layer_on.log | cut -c1-600
