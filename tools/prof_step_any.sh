#!/bin/bash
# rocprofv3 --kernel-trace --stats of one training step of ANY bench.py configuration:  tools/prof_step_any.sh <tag> <bench.py arguments...>
# per-step kernel table + the count of library GEMM kernels (Cijk_* / rocBLAS / hipBLASLt) -> gpurun_out/step_trace_<tag>.txt
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
OUT=gpurun_out/prof_step_$tag
rm -rf $OUT; mkdir -p $OUT
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-floor "$@" > $OUT/trace.log 2>&1
python - "$OUT" "gpurun_out/step_trace_$tag.txt" "$*" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
NS = 4  # bench.py --steps 2 --warmup 1 + its one extra untimed step (all kernel families timed)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
lib = [r["Name"] for r in rows if r["Name"].startswith("Cijk_") or "rocblas" in r["Name"].lower() or "hipblaslt" in r["Name"].lower()]
own = sum(float(r["TotalDurationNs"]) for r in rows if "anonymous namespace)::" in r["Name"] and "at::native" not in r["Name"])
out = [f"bench.py {sys.argv[3]}",
       f"total kernel time per step: {tot / NS / 1e6:.1f} ms ({NS} steps traced incl. warm-up and the extra instrumented step; model initialisation is in the totals)",
       f"library GEMM kernels in the trace (Cijk_* / rocBLAS / hipBLASLt): {len(lib)}" + ("" if not lib else " -- " + "; ".join(n[:60] for n in lib[:5])),
       f"share of the kernel time in this library's own kernels: {own / tot:.3f}"]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
    out.append(f"{float(r['TotalDurationNs']) / NS / 1e6:8.2f} ms/step {float(r['Percentage']):6.2f}% calls/step={int(r['Calls']) / NS:7.1f} avg_us={float(r['AverageNs']) / 1e3:9.1f}  {r['Name'][:150]}")
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[:16]))
PY
