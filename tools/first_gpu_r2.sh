#!/bin/bash
# Round 2, first GPU call: parity tests, same-box A/B of the scan kernels (round-1 build vs current), phase timing, bench.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
for v in r1 default r1 default; do
  if [ "$v" = default ]; then unset CADUCEUS_AMD_LIB; else export CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_$v.so; fi
  python tools/bwd_only_bench.py >> gpurun_out/ab_bwd.log 2>&1
done
unset CADUCEUS_AMD_LIB
cat gpurun_out/ab_bwd.log
CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_timing.so python tools/phase_timing.py > gpurun_out/phase_timing.log 2>&1
cat gpurun_out/phase_timing.log
python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -1 gpurun_out/bench.log
