#!/bin/bash
# kernel trace of the ORB quick bench; prints mean duration per (kernel, grid) of the last timed batches
NAME=$1; NP=${2:-64}
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$NAME -o kt -- python3 $R/tools/orb_quick_bench.py $NP > $R/gpurun_out/$NAME.log 2>&1 || true
cd $R
python3 - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/$NAME/*kernel_trace.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=(r['Kernel_Name'].replace('(anonymous namespace)::','')[:40], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    agg[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items()):
    v=v[len(v)//2:]
    print("%-42s grid=%-8s %-6s %-5s n=%3d mean=%8.1f us min=%8.1f"%(k[0],k[1],k[2],k[3],len(v),sum(v)/len(v),min(v)))
PY
