#!/bin/bash
# GPU box: same-box A/B of the whole training step (bench.py) over variant libraries.
# usage: tools/ab_bench_libs.sh <rounds> <variant> [<variant> ...]   ("default" = caduceus_amd/libcaduceus_hip.so)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rounds=$1; shift
: > gpurun_out/ab_bench_libs.log
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset CADUCEUS_AMD_LIB; else export CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_$v.so; fi
    timeout 120 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
o=j['roofline']['other_kernels_ms_per_step']
print('$v', round(j['value']), 'tok/s', round(j['ms_per_step'],2), 'ms/step | scan_fwd', round(j['roofline']['all']['scan_fwd']['avg_ms'],3), 'scan_bwd', round(j['roofline']['all']['scan_bwd']['avg_ms'],3), '|', ' '.join(f'{k}={v:.2f}' for k,v in o.items()))
" >> gpurun_out/ab_bench_libs.log
  done
done
unset CADUCEUS_AMD_LIB
cat gpurun_out/ab_bench_libs.log
