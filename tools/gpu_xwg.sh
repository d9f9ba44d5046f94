#!/bin/bash
# One short gpurun call: the cross-workgroup fold micro-benchmark (tools/ubench/xwg_fold.hip -- what an in-launch fold of the scan
# backward's dB/dC partial tiles would cost per chunk), its HBM-side counters, and the newest GPU test.
#   /usr/local/graft/bin/gpurun --timeout 480 -- 'bash tools/gpu_xwg.sh'
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/xwg_fold.txt
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o xwg_fold xwg_fold.hip ) 2>&1 | tail -3
X=tools/ubench/xwg_fold
: > $OUT
for cfg in "6000 64 2" "6000 64 1" "6000 64 3" "6000 8 2" "6000 16 2" "3000 64 2" "12000 64 2"; do
    timeout 60 $X $cfg >> $OUT 2>&1 || echo "xwg_fold $cfg: exit $?" >> $OUT
done
grep -c "wrong 0, spin timeouts 0" $OUT; grep -v "wrong 0, spin timeouts 0\|wrong -1" $OUT | head -20
# HBM-side traffic of the producer in each mode (separate passes; the dispatch order is compute-only, 0, 4, 1, 2 -- three times each)
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/xwg_pmc_$c -o xwg -- $X 6000 64 2 > gpurun_out/xwg_pmc_$c.log 2>&1
done
python - <<'PY'
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/xwg_pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(c, "no csv"); continue
    rows = [r for r in csv.DictReader(open(f[0])) if r.get("Counter_Name") == c]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    line = [f"{r['Kernel_Name'][:12]}:{float(r['Counter_Value']) / 1e6:.0f}" for r in rows]
    open("gpurun_out/xwg_fold.txt", "a").write(f"{c} (counter units x 1e6) per dispatch: " + " ".join(line) + "\n")
    print(c, " ".join(line)[:600])
PY
timeout 240 python -m pytest tests/test_kernels.py -m gpu -q -k "kernel_timer or lsplit" 2>&1 | tail -2
tail -40 $OUT
