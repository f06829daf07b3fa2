#!/usr/bin/env python3
"""Where a latency-bound chain spends its wall time (developer tool): reads a rocprofv3 --kernel-trace CSV, takes the launches of the
LAST `frames` repetitions of the chain (a repetition starts at each launch of `anchor`), and prints per kernel: launches per frame,
mean duration, and the idle gap in FRONT of it (start - previous end on the timeline), then the frame's totals.
  python tools/trace_gaps.py <kernel_trace.csv> [anchor-substring] [frames]"""
import csv
import sys
from collections import OrderedDict


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]


def main():
    path = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "orb_level_fused"
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(path))]
    rows.sort()
    # a frame starts at the first anchor launch after a non-anchor launch
    starts = [i for i, r in enumerate(rows) if anchor in r[2] and (i == 0 or anchor not in rows[i - 1][2])]
    if len(starts) < frames + 1:
        frames = max(1, len(starts) - 1)
    first = starts[-frames - 1]
    last = starts[-1]
    per = OrderedDict()
    busy = gap = 0
    for i in range(first, last):
        s, e, k = rows[i]
        g = max(0, s - rows[i - 1][1]) if i > first else 0
        d = per.setdefault(k, [0, 0, 0])
        d[0] += 1; d[1] += e - s; d[2] += g
        busy += e - s; gap += g
    wall = rows[last][0] - rows[first][0]
    print("%-46s %8s %10s %10s %10s" % ("kernel", "n/frame", "mean us", "us/frame", "gap us/frame"))
    for k, (n, d, g) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print("%-46s %8.2f %10.2f %10.2f %10.2f" % (k, n / frames, d / n / 1e3, d / frames / 1e3, g / frames / 1e3))
    print("frames %d: launches per frame %.1f, kernel time %.1f us, gaps %.1f us, wall %.1f us per frame"
          % (frames, (last - first) / frames, busy / frames / 1e3, gap / frames / 1e3, wall / frames / 1e3))


if __name__ == "__main__":
    main()
