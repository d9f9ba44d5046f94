#!/bin/bash
# GPU box: whole-step kernel trace (rocprofv3 --kernel-trace --stats) + per-kernel summary + the default bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 bash tools/prof_step.sh 2>&1 | tail -1 | cut -c1-200
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_step/trace/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    NS = 4  # bench.py --steps 2 --warmup 1 + its one extra untimed step (all kernel families timed)
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = [f"total kernel time per step: {tot / NS / 1e6:.1f} ms ({NS} steps traced incl. warm-up and the extra instrumented step; "
           "model initialisation is in the totals: fills, normal_)"]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
        out.append(f"{float(r['TotalDurationNs']) / NS / 1e6:8.2f} ms/step {float(r['Percentage']):6.2f}% calls/step={int(r['Calls']) / NS:7.1f} avg_us={float(r['AverageNs']) / 1e3:9.1f}  {r['Name'][:150]}")
    open("gpurun_out/step_trace.txt", "w").write("\n".join(out) + "\n")
    print("\n".join(out[:24]))
PY
timeout 400 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -1 gpurun_out/bench.log | cut -c1-400
