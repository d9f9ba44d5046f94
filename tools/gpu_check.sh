#!/bin/bash
# One gpurun call that validates a commit on the MI355X: the GPU parity tests, smoke(), the default bench line, and the same-box
# A/B of the two scan launches of a training step over any variant libraries given as arguments
# (built with caduceus_amd._build.build_hip(defines=..., out="caduceus_amd/libcaduceus_hip_<variant>.so")).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_check.sh [variant ...]'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1
grep -n "passed\|failed" gpurun_out/pytest_gpu.log | tail -3; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_gpu.log | head
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
if [ $# -gt 0 ]; then timeout 1200 bash tools/ab_train_scan.sh 2 default "$@" > /dev/null 2>&1; cat gpurun_out/ab_train_scan.log; fi
timeout 600 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -1 gpurun_out/bench.log | cut -c1-300
