#!/bin/bash
# rocprofv3 --kernel-trace --stats of one training step at the per-GPU shape of BASELINE configs[4] (Caduceus-PS d_model 512, seqlen 262144, 1 seq/GPU),
# bf16 and with the fp8 in_proj; per-step kernel table -> gpurun_out/step_trace_c4[_fp8].txt (what a library GEMM would show up in: a Cijk_* row)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for tag in "" "_fp8"; do
  OUT=gpurun_out/prof_step_c4$tag
  rm -rf $OUT; mkdir -p $OUT
  extra=""; [ "$tag" = "_fp8" ] && extra="--fp8-proj"
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --d-model 512 --seqlen 262144 --steps 2 --warmup 1 --cpu-sample 0 --no-floor $extra > $OUT/trace.log 2>&1
  tail -1 $OUT/trace.log | cut -c1-300
  python - "$OUT" "gpurun_out/step_trace_c4$tag.txt" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
NS = 4  # bench.py --steps 2 --warmup 1 + its one extra untimed step (all kernel families timed)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
lib = [r["Name"] for r in rows if r["Name"].startswith("Cijk_") or "rocblas" in r["Name"].lower() or "hipblaslt" in r["Name"].lower()]
out = [f"total kernel time per step: {tot / NS / 1e6:.1f} ms ({NS} steps traced incl. warm-up and the extra instrumented step; model initialisation is in the totals)",
       f"library GEMM kernels in the trace (Cijk_* / rocBLAS / hipBLASLt): {len(lib)}" + ("" if not lib else " -- " + "; ".join(n[:60] for n in lib[:5]))]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    out.append(f"{float(r['TotalDurationNs']) / NS / 1e6:8.2f} ms/step {float(r['Percentage']):6.2f}% calls/step={int(r['Calls']) / NS:7.1f} avg_us={float(r['AverageNs']) / 1e3:9.1f}  {r['Name'][:150]}")
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
PY
done
