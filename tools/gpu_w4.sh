#!/bin/bash
# GPU box: (1) the scan backward as TWO 4-wave workgroups per CU (-DSC_W_BWD=4, caduceus_amd/libcaduceus_hip_w4.so built by
# tools/build_variants.py w4=SC_W_BWD=4) against the production 8-wave workgroup: does the runtime place two per CU, same-box A/B of one
# mixer layer; (2) cad_gemm_stream with / without the XCD-aware item order (noxcd = -DGS_XCD_REMAP=0).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "" _w4; do
CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip$v.so python - <<'PY'
import ctypes, os
l = ctypes.CDLL(os.environ["CADUCEUS_AMD_LIB"]); o = (ctypes.c_int * 3)()
print("scan_bwd occupancy rc", l.cad_debug_scan_bwd_occupancy(o), "workgroups per CU", o[0], "LDS bytes", o[1], "waves per workgroup", o[2])
PY
done
for v in default noxcd default noxcd; do
  unset CADUCEUS_AMD_LIB; [ $v = default ] || export CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_$v.so
  echo "$v $(timeout 300 python tools/gemm_stream_bench.py 2>/dev/null | tail -1 | cut -c1-700)"
  echo "$v d512 $(timeout 300 python tools/gemm_stream_bench.py --d-model 512 --T 524288 --reps 8 2>/dev/null | tail -1 | cut -c1-700)"
done
unset CADUCEUS_AMD_LIB
LAYER_BENCH_ARGS="" bash tools/ab_layer.sh 2 default noxcd default,env:CADUCEUS_AMD_STREAM_FOLD=0 w4,env:CADUCEUS_AMD_STREAM_FOLD=0 | cut -c1-330
