#!/bin/bash
# rocprofv3 kernel stats of the tracking example on a generated sequence (developer tool)
R=$PWD
export TMPDIR=/tmp
python3 $R/examples/stereo_kitti.py /tmp/seqp --generate ${1:-20} > /dev/null 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/seqprof -o sq -- python3 $R/examples/stereo_kitti.py /tmp/seqp > $R/gpurun_out/seqprof.log 2>&1 || true
cd $R
tail -1 gpurun_out/seqprof.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/seqprof/sq_kernel_stats.csv')))
for r in rows[:16]:
    print("%-60s calls %4s avg %9.1f us total %8.1f us" % (r['Name'].replace('(anonymous namespace)::','')[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
