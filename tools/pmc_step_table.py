#!/usr/bin/env python3
"""Turns the three counter passes of tools/reproduce_profiles.sh into gpurun_out/<name>_traffic.json and <name>_valu_issue.json (per kernel,
summed over the kernel's launches of one timed step and averaged over the timed steps)."""
import collections
import csv
import glob
import json
import sys

name, steps, warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])


def images_per_step():
    """images one step of the profiled bench run handled (all lockstep groups), from the bench line in the pass's log"""
    try:
        for line in reversed(open("gpurun_out/%s_rd.log" % name).read().splitlines()):
            if line.startswith("{"):
                return int(json.loads(line)["config"]["images_per_step_per_gpu"])
    except (OSError, ValueError, KeyError):
        pass
    return 1024


IMAGES = images_per_step()
CLOCK_HZ = 2.4e9
SIMDS = 1024


def short(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]


def load(tag):
    f = glob.glob("gpurun_out/%s_%s/*counter_collection.csv" % (name, tag))
    per = collections.OrderedDict()          # kernel -> counter -> [values in dispatch order]
    if f:
        rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            per.setdefault(short(r["Kernel_Name"]), collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return per


def per_step(values):
    """sum per step over the timed steps: the kernel runs len / (steps + warm) times per step (set-up launches at the front dropped)"""
    n = len(values)
    if n < steps + warm:
        return None, 0
    lps = n // (steps + warm)
    v = values[n - steps * lps:]
    return sum(v) / steps, lps


rd, wr, sq = load("rd"), load("wr"), load("sq")
out = collections.OrderedDict()
for k, c in rd.items():
    n, lps = per_step(c.get("TCC_EA0_RDREQ", []))
    if n is None:
        continue
    n32, n64, n128 = (per_step(c.get(x, []))[0] or 0.0 for x in ("TCC_EA0_RDREQ_32B", "TCC_EA0_RDREQ_64B", "TCC_EA0_RDREQ_128B"))
    e = out.setdefault(k, collections.OrderedDict())
    e["launches_per_step"] = lps
    e["read_requests_per_step"] = {"all": n, "32B": n32, "64B": n64, "128B": n128}
    e["read_bytes_per_step"] = 32 * n32 + 64 * n64 + 128 * n128
    e["fetch_size_equivalent_bytes"] = 64 * n
for k, c in wr.items():
    n, lps = per_step(c.get("TCC_EA0_WRREQ", []))
    if n is None or k not in out:
        continue
    n64 = per_step(c.get("TCC_EA0_WRREQ_64B", []))[0] or 0.0
    out[k]["write_requests_per_step"] = {"all": n, "64B": n64}
    out[k]["write_bytes_per_step"] = 64 * n64 + 32 * (n - n64)
for k, e in out.items():
    e["hbm_bytes_per_step"] = e.get("read_bytes_per_step", 0.0) + e.get("write_bytes_per_step", 0.0)
json.dump({"images_per_launch": IMAGES, "images_per_launch_note": "images per STEP (all lockstep groups of the run); a launch of one of G groups handles 1 / G of it", "method": "tools/reproduce_profiles.sh: TCC_EA0_RDREQ size classes (32 / 64 / 128 B) and TCC_EA0_WRREQ(_64B), separate passes, "
           "bench.py --no-cpu --no-secondary --distinct 2; per kernel summed over its launches of one step, mean of %d timed steps" % steps,
           "kernels": out}, open("gpurun_out/%s_traffic.json" % name, "w"), indent=1)
iss = collections.OrderedDict()
for k, c in sq.items():
    v, lps = per_step(c.get("SQ_INSTS_VALU", []))
    if v is None:
        continue
    s = per_step(c.get("SQ_INSTS_SALU", []))[0] or 0.0
    w = per_step(c.get("SQ_WAVES", []))[0] or 0.0
    iss[k] = {"launches_per_step": lps, "waves_per_step": w, "valu_per_wave": v / w if w else None, "salu_per_wave": s / w if w else None,
              "valu_issue_ms_per_step": v * 4 / SIMDS / CLOCK_HZ * 1e3}
json.dump({"images_per_launch": IMAGES, "method": "tools/reproduce_profiles.sh: SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAVES; valu_issue_ms = instructions x 4 cycles / 1024 SIMDs / 2.4 GHz "
           "(the time the launch would take if VALU issue were the only limit)", "kernels": iss}, open("gpurun_out/%s_valu_issue.json" % name, "w"), indent=1)
for k, e in out.items():
    i = iss.get(k, {})
    print("%-22s x%d  read %8.1f MB (fetch-size equiv %8.1f)  write %8.1f MB   valu/wave %s  issue-ms %s" % (
        k[:22], e["launches_per_step"], e.get("read_bytes_per_step", 0) / 1e6, e["fetch_size_equivalent_bytes"] / 1e6, e.get("write_bytes_per_step", 0) / 1e6,
        "%.0f" % i["valu_per_wave"] if i.get("valu_per_wave") else "-", "%.3f" % i["valu_issue_ms_per_step"] if i else "-"))
