"""Union / sum of the kernel intervals of a rocprofv3 kernel trace (csv): how busy the GPU was and how much the streams overlapped.
  python tools/trace_overlap.py gpurun_out/<name>/<prefix>_kernel_trace.csv [skip_fraction]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
cut = t0 + (t1 - t0) * skip            # steady state: drop the warm-up part
iv = [x for x in iv if x[0] >= cut]
start, end = iv[0][0], max(e for _, e, _ in iv)
total = sum(e - s for s, e, _ in iv)
union, cur_s, cur_e = 0, None, None
for s, e, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("window %.3f ms: union of kernel intervals %.3f ms (%.1f %% busy), sum of kernel durations %.3f ms (mean concurrency %.2f)"
      % ((end - start) / 1e6, union / 1e6, 100.0 * union / (end - start), total / 1e6, total / union))
by = collections.defaultdict(float)
for s, e, n in iv:
    by[n.replace("(anonymous namespace)::", "").split("(")[0]] += e - s
for n, v in sorted(by.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-44s %8.3f ms  %5.1f %% of the sum" % (n[:44], v / 1e6, 100 * v / total))
