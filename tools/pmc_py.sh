#!/bin/bash
# PMC pass (counters only, with --kernel-trace) of an arbitrary python tool.  Usage: tools/pmc_py.sh <name> "<COUNTERS>" <script> [args]
NAME=$1; CNT=$2; shift 2
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $R/gpurun_out/$NAME -o pmc -- python3 $R/"$@" > $R/gpurun_out/$NAME.log 2>&1 || true
cd $R
python3 - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/$NAME/*counter_collection.csv')
if not f: print('no counter csv', glob.glob('gpurun_out/$NAME/*')); raise SystemExit
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'].replace('(anonymous namespace)::','')[:40]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    cnt[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    print(k, {c: round(x/cnt[(k,c)],1) for c,x in v.items()})
PY
