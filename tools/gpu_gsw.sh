#!/bin/bash
# GPU box: what binds cad_gemm_stream?  Timing builds (-DGS_WHATIF: wrong results) of the kernel with the multiplication, the operand
# streams or the LDS fragment reads compiled out, stand-alone at the weight-gradient / d(x2d) shapes of d_model 256 and 512.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp CADUCEUS_AMD_ALLOW_TIMING_BUILD=1
mkdir -p gpurun_out
: > gpurun_out/gemm_stream_whatif.txt
for v in default gsw1 gsw2 gsw4 gsw6; do
  unset CADUCEUS_AMD_LIB; [ $v = default ] || export CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_$v.so
  for args in "--d-model 256 --T 262144 --reps 12" "--d-model 512 --T 524288 --reps 6"; do
    echo "$v $args $(timeout 300 python tools/gemm_stream_bench.py $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v,4) for k,v in d.items() if k.endswith('_ms')})")" | tee -a gpurun_out/gemm_stream_whatif.txt
  done
done
