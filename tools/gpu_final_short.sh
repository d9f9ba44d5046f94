export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 240 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_final2.log 2>&1; grep -n "passed\|failed" gpurun_out/pytest_gpu_final2.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 120 python bench.py > gpurun_out/bench_final2.log 2>gpurun_out/bench_final2.err; tail -1 gpurun_out/bench_final2.log | cut -c1-300
timeout 120 bash tools/prof_step.sh 2>&1 | tail -1 | cut -c1-200
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_step/trace/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    NS = 4
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = [f"total kernel time per step: {tot / NS / 1e6:.1f} ms ({NS} steps traced incl. warm-up and the extra instrumented step; model initialisation is in the totals: fills, normal_)"]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
        out.append(f"{float(r['TotalDurationNs']) / NS / 1e6:8.2f} ms/step {float(r['Percentage']):6.2f}% calls/step={int(r['Calls']) / NS:7.1f} avg_us={float(r['AverageNs']) / 1e3:9.1f}  {r['Name'][:150]}")
    open("gpurun_out/step_trace_final2.txt", "w").write("\n".join(out) + "\n")
    print("\n".join(out[:8]))
PY
