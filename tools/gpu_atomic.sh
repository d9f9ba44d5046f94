#!/bin/bash
# GPU box: the dB / dC-by-atomics microbenchmark (tools/ubench/atomic_fold.hip): times, then FETCH_SIZE / WRITE_SIZE per mode in separate passes.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/atomic
mkdir -p $OUT
B=tools/ubench/atomic_fold
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $B $B.hip
timeout 300 $B 0 -1 3   > $OUT/times_iters0.txt 2>&1
timeout 300 $B 140 -1 3 > $OUT/times_iters140.txt 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$ctr -o af -- $B 0 -1 1 > $OUT/pmc_$ctr.log 2>&1
done
python - <<'PY' > gpurun_out/atomic/pmc_summary.txt 2>&1
import csv, glob, collections
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/atomic/pmc_{ctr}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]] += float(r["Counter_Value"])
        for k, v in acc.items():
            print(ctr, k, f"{v:.4g}")
PY
cat $OUT/times_iters0.txt $OUT/times_iters140.txt $OUT/pmc_summary.txt
