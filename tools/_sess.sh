mkdir -p gpurun_out
rm -f gpurun_out/tunable*.csv gpurun_out/tun.log
timeout 180 python tools/layer_bench.py 2>/dev/null | grep layer_ms >> gpurun_out/tun.log
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_results.csv PYTORCH_TUNABLEOP_VERBOSE=0
timeout 600 python tools/layer_bench.py 2>&1 | grep layer_ms >> gpurun_out/tun.log
export PYTORCH_TUNABLEOP_TUNING=0
timeout 180 python tools/layer_bench.py 2>&1 | grep layer_ms >> gpurun_out/tun.log
unset PYTORCH_TUNABLEOP_ENABLED PYTORCH_TUNABLEOP_TUNING PYTORCH_TUNABLEOP_FILENAME
timeout 180 python tools/layer_bench.py 2>/dev/null | grep layer_ms >> gpurun_out/tun.log
cut -c1-60 gpurun_out/tun.log; ls gpurun_out/tunable*; cat gpurun_out/tunableop_results*.csv | head -30
