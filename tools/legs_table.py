#!/usr/bin/env python3
"""Turns the passes of tools/reproduce_profiles.sh into gpurun_out/<name>_<leg>_kernel_stats.csv (rocprofv3 --stats table of the leg) and
gpurun_out/<name>_legs_pmc.json (per leg and kernel: launches, mean duration, HBM read / write bytes per launch from the request
size classes, VALU / SALU instructions per wave, matrix-core busy cycles per launch).  bench.py reads the committed copy
(profiles/<round>_legs_pmc.json) into metric_ba.roofline.traffic and the a10 / a14 / a15 rooflines."""
import collections
import csv
import glob
import json
import shutil
import sys

name, legs = sys.argv[1], sys.argv[2:]


def short(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def counters(leg, tag):
    f = glob.glob("gpurun_out/%s/%s_%s/*counter_collection.csv" % (name, leg, tag))
    per = collections.OrderedDict()
    if f:
        for r in csv.DictReader(open(f[0])):
            per.setdefault(short(r["Kernel_Name"]), collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


def mean(v):
    return sum(v) / len(v) if v else 0.0


out = collections.OrderedDict()
for leg in legs:
    st = glob.glob("gpurun_out/%s/%s_kt/*kernel_stats.csv" % (name, leg))
    if st:
        shutil.copy(st[0], "gpurun_out/%s_%s_kernel_stats.csv" % (name, leg))
    kt = glob.glob("gpurun_out/%s/%s_kt/*kernel_trace.csv" % (name, leg))
    dur = collections.OrderedDict()
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur.setdefault(short(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rd, wr, sq = counters(leg, "rd"), counters(leg, "wr"), counters(leg, "sq")
    e = out.setdefault(leg, collections.OrderedDict())
    for k, d in dur.items():
        x = e.setdefault(k, collections.OrderedDict())
        x["launches"] = len(d)
        x["mean_us"] = round(mean(d) / 1e3, 3)
        x["total_ms"] = round(sum(d) / 1e6, 4)
        c = rd.get(k)
        if c:
            x["read_bytes_per_launch"] = 32 * mean(c["TCC_EA0_RDREQ_32B"]) + 64 * mean(c["TCC_EA0_RDREQ_64B"]) + 128 * mean(c["TCC_EA0_RDREQ_128B"])
        c = wr.get(k)
        if c:
            n, n64 = mean(c["TCC_EA0_WRREQ"]), mean(c["TCC_EA0_WRREQ_64B"])
            x["write_bytes_per_launch"] = 64 * n64 + 32 * (n - n64)
        c = sq.get(k)
        if c:
            w = mean(c["SQ_WAVES"])
            x["waves_per_launch"] = w
            x["valu_per_wave"] = round(mean(c["SQ_INSTS_VALU"]) / w, 1) if w else None
            x["salu_per_wave"] = round(mean(c["SQ_INSTS_SALU"]) / w, 1) if w else None
            x["valu_per_launch"] = mean(c["SQ_INSTS_VALU"])
            x["mfma_busy_cycles_per_launch"] = mean(c.get("SQ_VALU_MFMA_BUSY_CYCLES", []))
            x["sq_busy_cycles_per_launch"] = mean(c.get("SQ_BUSY_CYCLES", []))
    tot = sum(x["total_ms"] for x in e.values()) or 1.0
    print("== %s" % leg)
    for k, x in sorted(e.items(), key=lambda kv: -kv[1]["total_ms"]):
        print("%-28s n=%5d mean %9.1f us  %5.1f%%  rd %9.1f KB wr %9.1f KB  valu/wave %s  mfma busy %s" % (
            k[:28], x["launches"], x["mean_us"], 100 * x["total_ms"] / tot, x.get("read_bytes_per_launch", 0) / 1e3,
            x.get("write_bytes_per_launch", 0) / 1e3, x.get("valu_per_wave"), x.get("mfma_busy_cycles_per_launch")))
json.dump({"method": "tools/reproduce_profiles.sh: rocprofv3 --kernel-trace (durations) and three separate --pmc passes (TCC_EA0_RDREQ size classes, "
           "TCC_EA0_WRREQ(_64B), SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAVES / SQ_VALU_MFMA_BUSY_CYCLES) of the leg's tool script; every launch of "
           "the process, means per launch", "legs": out}, open("gpurun_out/%s_legs_pmc.json" % name, "w"), indent=1)
