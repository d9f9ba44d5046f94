"""GPU box, after tools/prof_scan.sh: condenses the rocprofv3 counter CSVs of the scan micro-benchmark into the JSON that bench.py
quotes as `roofline.traffic` (bench.SCAN_PMC_FILE), stamped with cad_version() of the library that was profiled.
Per-dispatch averages of the two-set production launches only (selected by their grid size); FETCH_SIZE x 2 and the KiB unit
per the gfx950 notes of MI355X_MICROARCH.md."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import _build, _lib  # noqa: E402

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
out_path = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/scan_pmc.json"
E, ROWS, L, N = 512, 2, 131072, 16
two_set_threads = (E // 8) * 512 * ROWS * 2  # grid of the two-set launch, in threads

acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        kind = "scan_fwd" if "scan_fwd_kernel" in name else "scan_bwd" if "scan_bwd_kernel" in name else None
        if kind is None:
            continue
        grid = int(float(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
        if grid and grid != two_set_threads:
            continue
        acc[kind][r["Counter_Name"]].append(float(r["Counter_Value"] or 0))
res = {"source": "rocprofv3 --pmc passes of tools/prof_scan.sh over tools/layer_bench.py (one production mixer layer) (FETCH_SIZE, WRITE_SIZE and two SQ "
                 "groups, each in its own pass, no tracing combined); per-dispatch averages of the two-set production launches; "
                 "FETCH_SIZE x 2 (gfx950 correction), both size counters reported in KiB",
       "lib_version": _lib.version(), "scan_src": _build.scan_source_hash(),
       "shape": {"E": E, "rows": ROWS, "L": L, "N": N, "dtype": "bf16", "sets": 2}, "sq": {}}
for kind, ctrs in acc.items():
    avg = {c: sum(v) / len(v) for c, v in ctrs.items()}
    res[kind] = {"fetch_bytes": avg.get("FETCH_SIZE", 0.0) * 1024 * 2, "write_bytes": avg.get("WRITE_SIZE", 0.0) * 1024,
                 "dispatches": {c: len(v) for c, v in ctrs.items()}}
    res["sq"][kind] = {c: v for c, v in avg.items() if c.startswith("SQ_")}
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("lib_version", "scan_fwd", "scan_bwd") if k in res}))
