#!/bin/bash
# kernel trace of an arbitrary python tool: tools/trace_py.sh <name> <script> [args]; prints mean duration per kernel
NAME=$1; shift
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$NAME -o kt -- python3 $R/"$@" > $R/gpurun_out/$NAME.log 2>&1 || true
cd $R
python3 - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/$NAME/*kernel_trace.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].replace('(anonymous namespace)::','')[:44]
    agg[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot=sum(sum(v) for v in agg.values())
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    print("%-46s n=%5d total=%9.1f us mean=%8.1f max=%8.1f  %4.1f%%"%(k,len(v),sum(v),sum(v)/len(v),max(v),100*sum(v)/tot))
PY
tail -3 gpurun_out/$NAME.log
