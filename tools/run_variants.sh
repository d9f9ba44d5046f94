#!/bin/bash
# GPU box: instruction microbench, scan microbench for the default library and tuning variants, quick parity subset.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
./tools/ubench/ubench > gpurun_out/ubench.log 2>&1
python -m pytest tests -m gpu -x -q -k "scan or equivariance" > gpurun_out/pytest_scan.log 2>&1; tail -3 gpurun_out/pytest_scan.log
echo "default" > gpurun_out/variants.log
python tools/scan_bench.py --reps 5 --only-scan >> gpurun_out/variants.log 2>&1
for v in "$@"; do
  echo "$v" >> gpurun_out/variants.log
  CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_$v.so python tools/scan_bench.py --reps 5 --only-scan >> gpurun_out/variants.log 2>&1
done
cat gpurun_out/variants.log
