#!/usr/bin/env python3
"""Developer tool: what the lockstep groups' overlap does to every kernel of the headline loop.

Reads the two kernel traces tools/reproduce_profiles.sh leaves (<dir>/orb_kernel_trace.csv: the default run, several lockstep groups on
their own streams; <dir>_1group/...: the same sequences in ONE group, the kernels alone) and prints, over the timed steps of
each run: how much of the wall time 0 / 1 / 2 / 3 ... kernels were running at once, and per kernel its share of the wall time summed over
the streams, its mean duration under the overlap and the mean duration of the single-group launch divided by the number of groups (what a
launch of one group's size would take alone if the kernel scaled with its batch - latency-bound kernels do not, their ratio overstates).
  python tools/overlap_table.py <dir> <groups> [steps warmup]  >  profiles/rNN_overlap_table.txt"""
import collections
import csv
import sys


def load(path, steps=20, warmup=3):
    ev = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:30]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    ev.sort()
    # the timed loop: from the end of the last warm-up step to the end of the last step (trk_stamp_overflow closes a group's camera chain;
    # bench.py defaults: 20 timed steps behind 3 warm-up steps)
    marks = [e for _, e, n in ev if n.startswith("trk_stamp_overflow")]
    per = len(marks) // (steps + warmup)
    lo, hi = marks[warmup * per - 1], marks[(steps + warmup) * per - 1]
    agg, cnt = collections.Counter(), collections.Counter()
    pts = []
    for s, e, n in ev:
        if e < lo or s > hi:
            continue
        pts += [(max(s, lo), 1), (min(e, hi), -1)]
        if s >= lo and e <= hi:
            agg[n] += e - s
            cnt[n] += 1
    pts.sort()
    cur, last, hist = 0, lo, collections.Counter()
    for t, d in pts:
        hist[cur] += t - last
        last = t
        cur += d
    return agg, cnt, hi - lo, hist


def main():
    name, groups = sys.argv[1], int(sys.argv[2])
    steps, warmup = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (20, 3)
    a3, c3, w3, h3 = load("%s/orb_kernel_trace.csv" % name, steps, warmup)
    a1, c1, w1, _ = load("%s_1group/orb_kernel_trace.csv" % name, steps, warmup)
    tot = sum(h3.values())
    print("%d lockstep groups: share of the wall time with k kernels running: %s" % (groups, {k: round(v / tot, 3) for k, v in sorted(h3.items())}))
    print("kernel time summed over the streams / wall time: %.2f   (one group: %.2f)" % (sum(a3.values()) / w3, sum(a1.values()) / w1))
    print("%-32s %14s %16s %22s %6s" % ("kernel", "share of wall", "mean us, overlap", "mean us alone / groups", "ratio"))
    for n, v in a3.most_common():
        m3 = v / c3[n] / 1e3
        m1 = a1[n] / max(c1[n], 1) / 1e3 / groups
        print("%-32s %14.3f %16.1f %22.1f %6.1f" % (n, v / w3, m3, m1, m3 / m1 if m1 > 0 else 0.0))


if __name__ == "__main__":
    main()
