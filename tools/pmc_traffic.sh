#!/bin/bash
# HBM traffic of the ORB kernels from PMC counters (two separate passes: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2),
# run on the GPU box.  Writes gpurun_out/<name>_traffic.json with per-kernel per-launch bytes.
# gfx950 correction (/opt/skills/guides/MI355X_MICROARCH.md, section HBM): FETCH_SIZE counts 64 B per 128-B request of a wide
# coalesced stream, so the raw value is reported next to the doubled one; WRITE_SIZE is uncalibrated.  Units: KiB.
NAME=${1:-traffic}; NP=${2:-64}
R=$PWD
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/${NAME}_$C -o pmc -- python3 $R/tools/orb_quick_bench.py $NP > $R/gpurun_out/${NAME}_$C.log 2>&1 || true
  cd $R
done
python3 - <<PY
import csv,glob,collections,json
out={}
for C in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob('gpurun_out/${NAME}_%s/*counter_collection.csv'%C)
    if not f: continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name']!=C: continue
        agg[r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0].replace('void ','')].append(float(r['Counter_Value']))
    for k,v in agg.items():
        v=v[len(v)//2:]
        out.setdefault(k,{})[C+"_KiB_per_launch"]=sum(v)/len(v)
for k,v in out.items():
    f=v.get("FETCH_SIZE_KiB_per_launch",0); w=v.get("WRITE_SIZE_KiB_per_launch",0)
    v["hbm_bytes_per_launch_raw"]=(f+w)*1024
    v["hbm_bytes_per_launch_fetch_doubled"]=(2*f+w)*1024
json.dump({"images_per_launch": 2*$NP, "kernels": out}, open('gpurun_out/${NAME}_traffic.json','w'), indent=1)
print(json.dumps(out, indent=1))
PY
