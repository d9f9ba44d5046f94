"""Per-kernel table from tools/prof_step_pmc.sh: MFMA instructions and busy cycles, MFMA utilisation, HBM-side bytes, duration.
Counter values are summed over the dispatches of a kernel and divided by their number (per-launch averages).
FETCH_SIZE is in KiB and counts 64-byte requests as 32 on gfx950 (x2, /opt/skills/guides/MI355X_MICROARCH.md); WRITE_SIZE in KiB."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(root + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "anonymous namespace" not in k and not k.startswith("Cijk"):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
dur = {}
for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = float(r["AverageNs"]) / 1e3
print(f"{'kernel':78s} {'n':>3s} {'us':>8s} {'MFMA instr':>11s} {'MFMA busy':>11s} {'util':>6s} {'fetch MB':>9s} {'write MB':>9s} {'TB/s':>5s}")
for k in sorted(acc, key=lambda k: -dur.get(k, 0)):
    c = acc[k]
    per = lambda name: c[name] / max(1, len(ndisp[k][name])) if name in c else float("nan")
    n = max(len(v) for v in ndisp[k].values())
    gui = per("GRBM_GUI_ACTIVE")
    busy = per("SQ_VALU_MFMA_BUSY_CYCLES")
    util = busy / (gui / 8 * 1024) if gui == gui and gui > 0 else float("nan")  # GUI_ACTIVE summed over 8 XCDs; 1024 SIMDs
    fetch = per("FETCH_SIZE") * 1024 * 2 / 1e6
    write = per("WRITE_SIZE") * 1024 / 1e6
    us = dur.get(k, float("nan"))
    tbs = (fetch + write) / us if us == us and us > 0 else float("nan")
    name = k.replace("void (anonymous namespace)::", "")[:78]
    print(f"{name:78s} {n:3d} {us:8.1f} {per('SQ_INSTS_MFMA'):11.0f} {busy:11.0f} {util:6.1%} {fetch:9.1f} {write:9.1f} {tbs:5.2f}")
