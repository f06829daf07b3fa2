#!/usr/bin/env python3
"""Per-kernel statistics of the TIMED launches of a rocprofv3 --kernel-trace of bench.py (developer tool).

rocprofv3 --stats averages every launch of the process, warm-up included, and the first launch of a kernel pays the code-object
load (12 ms for orb_fast_cells against 2 ms afterwards).  This script reads the kernel trace of the same run, drops the launches
that belong to the warm-up steps (the first warmup / (warmup + steps) of each kernel's launches) and writes the table in the
--stats column layout, so that the figure bench.py measures with HIP events over its timed region has its counterpart.
  python tools/trace_stats.py <kernel_trace.csv> <steps> <warmup> > <out.csv>"""
import csv
import sys
from collections import OrderedDict


def main():
    path, steps, warmup = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    per = OrderedDict()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        # torch's own kernels (the bench's set-up indexing: 20 launches that land in no timed step) and the runtime's fills are not the step's
        if name.startswith("at::") or name.startswith("void at::") or "elementwise" in name:
            continue
        per.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = []
    for name, d in per.items():
        if len(d) < steps + warmup:             # not a per-step kernel of the timed loop (set-up work): left out
            continue
        d = d[len(d) % (steps + warmup):]       # launches before the loop (handle creation runs some kernels once)
        per_step = len(d) // (steps + warmup)
        t = d[warmup * per_step:]
        mean = sum(t) / len(t)
        var = sum((x - mean) ** 2 for x in t) / len(t)
        rows.append((sum(t), name, len(t), mean, min(t), max(t), var ** 0.5))
    total = sum(r[0] for r in rows) or 1
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for tot, name, n, mean, mn, mx, sd in sorted(rows, reverse=True):
        w.writerow([name, n, tot, round(mean, 3), round(100.0 * tot / total, 2), mn, mx, round(sd, 3)])


if __name__ == "__main__":
    main()
