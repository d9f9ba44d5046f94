#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python tools/grad_spread.py > gpurun_out/s8_grad_spread.json 2>/dev/null; python - <<'PY'
import json
d=json.load(open("gpurun_out/s8_grad_spread.json"))
print("loss_equal", d["loss_equal"]); print({k:(v["bit_equal"], float("%.2g"%v["rel_to_max"])) for k,v in d["params"].items()})
PY
timeout 900 python -m pytest tests/test_configs.py tests/test_fp8.py tests/test_kernels.py -m gpu -x -q -k "config4 or fp8 or reduce_partials or add_norm" > gpurun_out/s8_pytest.log 2>&1; tail -3 gpurun_out/s8_pytest.log; grep -n "config4 one-layer" gpurun_out/s8_pytest.log | cut -c1-600
timeout 400 python bench.py > gpurun_out/s8_bench.log 2> gpurun_out/s8_bench.err; tail -1 gpurun_out/s8_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step')}, r['frac'], r['avg_launch_ms'], r.get('arithmetic_floor'), d['cpu_baseline']['value'], d['cpu_baseline'].get('spread'), d['cpu_baseline']['cores'])"
