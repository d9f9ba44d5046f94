#!/bin/bash
# GPU box: the concurrent dB / dC fold -- parity + stress tests, same-box A/B (fold on the second stream vs fold kernel behind the scan;
# write-through vs plain slot stores), and a kernel trace of one layer that shows whether the two kernels really overlap.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/fold
timeout 900 python -m pytest tests/test_fold_stream.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python tools/layer_bench.py --ab _STREAM_FOLD --reps 6 --rounds 4 2>&1 | tail -1
LAYER_BENCH_ARGS="" bash tools/ab_layer.sh 3 default env:CADUCEUS_AMD_STREAM_FOLD=0 nowt,env:CADUCEUS_AMD_STREAM_FOLD=0 | cut -c1-400
timeout 300 python tools/layer_bench.py --d-model 512 --seqlen 262144 --ab _STREAM_FOLD --reps 3 --rounds 3 2>&1 | tail -1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fold/trace -o layer -- python tools/layer_bench.py --reps 2 > gpurun_out/fold/trace.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/fold/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sc = [r for r in rows if "scan_bwd_kernel" in r["Kernel_Name"]]
fo = [r for r in rows if "fold_stream_kernel" in r["Kernel_Name"]]
print("scan_bwd launches", len(sc), "fold launches", len(fo))
for s in sc[-2:]:
    s0, s1 = int(s["Start_Timestamp"]), int(s["End_Timestamp"])
    near = [r for r in fo if abs(int(r["Start_Timestamp"]) - s0) < 20e6]
    print(f"scan_bwd {((s1 - s0) / 1e3):9.1f} us")
    for r in near:
        f0, f1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"   fold: starts {((f0 - s0) / 1e3):8.1f} us after the scan starts, runs {((f1 - f0) / 1e3):8.1f} us, ends {((f1 - s1) / 1e3):8.1f} us after the scan ends")
PY
