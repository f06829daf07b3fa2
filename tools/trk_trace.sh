#!/bin/bash
# rocprofv3 kernel trace + stats of the lockstep tracker bench (run on the GPU box): per-kernel mean duration of a step's launches
# usage: tools/trk_trace.sh NAME S FRAMES GROUPS
NAME=${1:-trk_kt}; S=${2:-128}; N=${3:-8}; G=${4:-1}
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$NAME -o trk -- python3 $R/tools/track_bench.py $S $N $G > $R/gpurun_out/$NAME.log 2>&1 || true
cd $R
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/$NAME/**/*kernel_stats.csv', recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    print("%-64s calls %5s avg %10.1f us total %8.2f ms %5.1f%%" % (r["Name"].replace("(anonymous namespace)::","")[:64], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, 100*float(r["TotalDurationNs"])/tot))
PY
