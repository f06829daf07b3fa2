#!/bin/bash
# HBM traffic of the ORB kernels from PMC counters (two separate passes: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2), run on the
# GPU box.  Writes gpurun_out/<name>_traffic.json with per-kernel per-launch bytes.
# Corrections (/opt/skills/guides/MI355X_MICROARCH.md, section HBM): FETCH_SIZE is calibrated there only for 16 B / lane streaming
# reads (it reports half of the bytes: doubled); "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte
# count in your own access pattern".  The same two passes therefore also run the library's known-byte streaming kernels
# (tools/pmc_calib.py: 1 GiB read at 16 B and at 4 B per lane, written at 4 B and at 16 B per lane) and the factors
# known bytes / counter are applied per access width: the ORB kernels load and store dwords.  Units of the counters: KiB.
NAME=${1:-traffic}; NP=${2:-64}
R=$PWD
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/${NAME}_$C -o pmc -- python3 $R/tools/orb_quick_bench.py $NP > $R/gpurun_out/${NAME}_$C.log 2>&1 || true
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/${NAME}_cal_$C -o pmc -- python3 $R/tools/pmc_calib.py > $R/gpurun_out/${NAME}_cal_$C.log 2>&1 || true
  cd $R
done
python3 - <<PY
import csv,glob,collections,json
def short(k): return k.replace('(anonymous namespace)::','').split('(')[0].replace('void ','')
def per_kernel(d, C):
    f=glob.glob('gpurun_out/%s/*counter_collection.csv' % d)
    agg=collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if r['Counter_Name']==C: agg[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return agg
GIB=float(1<<30)
cal={}
for C in ("FETCH_SIZE","WRITE_SIZE"):
    for k,v in per_kernel('${NAME}_cal_'+C, C).items():
        v=v[1:] if len(v)>1 else v          # first launch of each mode: cold
        cal[(C,k)]=sum(v)/len(v)*1024
# factors: known bytes / counted bytes
fac={"read16": GIB/cal.get(("FETCH_SIZE","traffic_read<HIP_vector_type<unsigned int, 4u> >"),float('nan')),
     "read4": GIB/cal.get(("FETCH_SIZE","traffic_read<unsigned int>"),float('nan')),
     "write4": GIB/cal.get(("WRITE_SIZE","traffic_write<unsigned int>"),float('nan')),
     "write16": GIB/cal.get(("WRITE_SIZE","traffic_write<HIP_vector_type<unsigned int, 4u> >"),float('nan'))}
out={}
for C in ("FETCH_SIZE","WRITE_SIZE"):
    for k,v in per_kernel('${NAME}_'+C, C).items():
        v=v[len(v)//2:]
        out.setdefault(k,{})[C+"_KiB_per_launch"]=sum(v)/len(v)
for k,v in out.items():
    f=v.get("FETCH_SIZE_KiB_per_launch",0)*1024; w=v.get("WRITE_SIZE_KiB_per_launch",0)*1024
    v["hbm_bytes_per_launch_raw"]=f+w
    v["hbm_bytes_per_launch_fetch_doubled"]=2*f+w
    v["hbm_bytes_per_launch"]=f*fac["read4"]+w*fac["write4"]      # the ORB kernels load and store dwords
json.dump({"images_per_launch": 2*$NP, "calibration": {"bytes_per_launch": GIB, "counter_bytes": {"%s %s"%k: v for k,v in cal.items()}, "factors_known_over_counted": fac,
           "applied": "hbm_bytes_per_launch = FETCH_SIZE x read4 + WRITE_SIZE x write4 (dword loads / stores)"}, "kernels": out}, open('gpurun_out/${NAME}_traffic.json','w'), indent=1)
print(json.dumps({"factors": fac, "kernels": out}, indent=1))
PY
