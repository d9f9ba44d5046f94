mkdir -p gpurun_out
timeout 600 python tools/find_fills.py > gpurun_out/find_fills.log 2>&1; tail -70 gpurun_out/find_fills.log
