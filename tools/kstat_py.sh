#!/bin/bash
# kernel trace of an arbitrary python tool; prints the mean duration per kernel over the second half of its launches
# usage: tools/kstat_py.sh <name> <script> [args]     (PS_LIB_PATH selects an experiment build)
NAME=$1; shift
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$NAME -o kt -- python3 $R/"$@" > $R/gpurun_out/$NAME.log 2>&1 || true
cd $R
python3 - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/$NAME/*kernel_trace.csv')[0]
agg=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:40]
    agg.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot=0
for k,v in agg.items():
    h=v[len(v)//2:]
    per_call = len(v)
    print("%-42s n=%4d mean=%8.1f us min=%8.1f"%(k,len(v),sum(h)/len(h),min(h)))
PY
