#!/bin/bash
# GPU session 2 of round 5: lane states (forward writes the state entering every 8-position segment, the bf16 backward starts from it)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "from caduceus_amd import _lib; print(_lib.version())" > gpurun_out/s2_version.txt 2>&1
timeout 1500 python -m pytest tests/test_kernels.py tests/test_proj.py tests/test_equivariance.py tests/test_configs.py tests/test_full_size.py -m gpu -x -q \
  -k "scan or lean or production_scans or mirror or mixer_layer or config2 or L131072 or full_length" > gpurun_out/s2_pytest.log 2>&1
tail -3 gpurun_out/s2_pytest.log
bash tools/ab_layer.sh 3 base env:CADUCEUS_AMD_LANE_STATES=0 default nt_proj_x_loads > gpurun_out/s2_ab.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/s2_ab.log"):
    if l.startswith("{"):
        r = json.loads(l)
        print(f"{r['lib'][-34:]:34s} layer {r['layer_ms']:7.3f} fwd {r['scan_fwd_ms']:7.4f} bwd {r['scan_bwd_ms']:7.4f} proj {r.get('proj_ms', 0):7.4f} conv {r.get('conv_fwd_ms',0):.4f}/{r.get('conv_bwd_ms',0):.4f}")
PY
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/s2_bench.log 2>&1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"kernel_over_floor": {[^}]*}' gpurun_out/s2_bench.log | head
