"""Kernel trace with kernels that run SIDE BY SIDE (the dB / dC fold on its second stream): the sum of the kernel durations is no longer the
time the GPU was busy.  From a rocprofv3 --kernel-trace CSV: the union of all kernel intervals (GPU-busy time), the plain sum, and for
every kernel name how much of its duration was covered by another kernel (overlapped) and how much it alone kept the GPU busy (exposed).
    python tools/trace_overlap.py <dir or kernel_trace.csv> [steps]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(path):
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    return rows


def main():
    rows = load(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    # sweep: at every instant, the set of running kernels; time with exactly one running kernel is that kernel's exposed time
    ev = []
    for i, (a, b, _n) in enumerate(rows):
        ev.append((a, 1, i))
        ev.append((b, 0, i))
    ev.sort()
    running = set()
    busy = 0
    exposed = defaultdict(int)
    last = None
    for t, kind, i in ev:
        if running and last is not None and t > last:
            busy += t - last
            if len(running) == 1:
                exposed[rows[next(iter(running))][2]] += t - last
        last = t
        if kind == 1:
            running.add(i)
        else:
            running.discard(i)
    dur = defaultdict(int)
    cnt = defaultdict(int)
    for a, b, n in rows:
        dur[n] += b - a
        cnt[n] += 1
    tot = sum(dur.values())
    print(f"GPU busy (union of kernel intervals) per step: {busy / steps / 1e6:.2f} ms;  sum of kernel durations per step: {tot / steps / 1e6:.2f} ms "
          f"({steps:g} steps traced incl. warm-up / instrumented steps)")
    print(f"{'exposed ms/step':>16} {'duration ms/step':>17} {'overlapped':>11} {'calls/step':>11} {'avg us':>9}  kernel")
    for n in sorted(dur, key=lambda k: -dur[k])[:45]:
        print(f"{exposed[n] / steps / 1e6:16.2f} {dur[n] / steps / 1e6:17.2f} {100.0 * (1 - exposed[n] / max(1, dur[n])):10.1f}% "
              f"{cnt[n] / steps:11.1f} {dur[n] / cnt[n] / 1e3:9.1f}  {n[:130]}")


if __name__ == "__main__":
    main()
