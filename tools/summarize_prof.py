"""Condense rocprofv3 CSV output (kernel stats + PMC counter collection) into a short text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
out = []
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
    out.append(f"## {f}")
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        out.append("  ".join(f"{k}={r[k]}" for k in r))
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    out.append(f"## {f}")
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?")[:70]
        acc[k][r.get("Counter_Name", "?")] += float(r.get("Counter_Value", 0) or 0)
        cnt[(k, r.get("Counter_Name", "?"))] += 1
    for k in acc:
        out.append(f"kernel {k}")
        for c, v in sorted(acc[k].items()):
            n = cnt[(k, c)]
            out.append(f"    {c}: total={v:.6g} dispatches={n} per_dispatch={v / max(n, 1):.6g}")
print("\n".join(out))
