#!/bin/bash
# PMC counters of the tracking example's kernels (developer tool).  Usage: tools/pmc_seq.sh <name> "<COUNTERS>"
NAME=$1; CNT=$2
R=$PWD
export TMPDIR=/tmp
python3 $R/examples/stereo_kitti.py /tmp/seqp --generate 8 > /dev/null 2>&1
cd /tmp
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $R/gpurun_out/$NAME -o pmc -- python3 $R/examples/stereo_kitti.py /tmp/seqp > $R/gpurun_out/$NAME.log 2>&1 || true
cd $R
python3 - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/$NAME/*counter_collection.csv')
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'][:40]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    if 'pj_resolve' in k or 'pose_lm' in k: print(k, {c: round(x/cnt[(k,c)],1) for c,x in v.items()})
PY
