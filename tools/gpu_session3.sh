#!/bin/bash
# round-3 GPU session 3: regression check vs the round-2 kernels, parity of the fp8 / L-split / configs[4] paths, bench lines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 400 bash tools/ab_layer.sh 2 r2 default
timeout 900 python -m pytest tests/test_proj.py tests/test_fp8.py tests/test_kernels.py -m gpu -x -q > gpurun_out/pytest_r3c.log 2>&1; grep -n "passed\|failed" gpurun_out/pytest_r3c.log | tail -2
timeout 1200 python -m pytest tests/test_configs.py -m gpu -q -s > gpurun_out/pytest_configs_r3c.log 2>&1; grep -n "config4 fp8\|passed\|failed\|Error\|assert" gpurun_out/pytest_configs_r3c.log | cut -c1-400 | tail -12
for k in 1 2 4; do
  CADUCEUS_AMD_LSPLIT=$k timeout 300 python bench.py --model ph --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/bench_ph_lsplit$k.log 2>gpurun_out/bench_ph_lsplit$k.err; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_ph_lsplit$k.log").read().strip().splitlines()[-1])
    print("ph lsplit=$k", round(d["value"]), "tok/s", round(d["ms_per_step"], 2), "ms", {k2: round(v["avg_ms"], 3) for k2, v in d["roofline"]["all"].items()}, "frac", round(d["roofline"]["frac"], 4))
except Exception as e:
    print("ph lsplit=$k failed", e)
PY
done
timeout 400 python bench.py --d-model 512 --seqlen 262144 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/bench_c4_bf16.log 2>gpurun_out/bench_c4_bf16.err; tail -1 gpurun_out/bench_c4_bf16.log | cut -c1-330
timeout 400 python bench.py --d-model 512 --seqlen 262144 --steps 3 --warmup 1 --cpu-sample 0 --fp8-proj > gpurun_out/bench_c4_fp8.log 2>gpurun_out/bench_c4_fp8.err; tail -1 gpurun_out/bench_c4_fp8.log | cut -c1-330; python - <<PY
import json
for f in ("gpurun_out/bench_c4_bf16.log", "gpurun_out/bench_c4_fp8.log"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, json.dumps(d["roofline"]["projections"])[:900])
    except Exception as e:
        print(f, "failed", e)
PY
timeout 400 python bench.py > gpurun_out/bench_r3c.log 2>gpurun_out/bench_r3c.err; tail -1 gpurun_out/bench_r3c.log | cut -c1-330
