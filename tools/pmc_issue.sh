#!/bin/bash
# VALU issue occupancy of the ORB kernels from PMC counters (own pass, --kernel-trace only), run on the GPU box.
# Writes gpurun_out/<name>_issue.json: per kernel, VALU instructions per wave, waves per launch, mean duration, and the share of
# the duration that waves x instructions x 4 cycles (one wave64 VALU instruction occupies a SIMD for 4 cycles) / (1024 SIMDs x
# shader clock) accounts for.
NAME=${1:-issue}; NP=${2:-64}; CLK_GHZ=${3:-2.4}
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/${NAME}_pmc -o pmc -- python3 $R/tools/orb_quick_bench.py $NP > $R/gpurun_out/${NAME}_pmc.log 2>&1 || true
cd $R
python3 - <<PY
import csv,glob,collections,json
def short(k): return k.replace('(anonymous namespace)::','').split('(')[0].replace('void ','')
cnt=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob('gpurun_out/${NAME}_pmc/*counter_collection.csv')[0])):
    cnt[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
dur=collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob('gpurun_out/${NAME}_pmc/*kernel_trace.csv')[0])):
    dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
out={}
for k,c in cnt.items():
    if not k.startswith('orb_') and not k.startswith('st_'): continue
    half=lambda v: v[len(v)//2:]
    m=lambda v: sum(half(v))/max(len(half(v)),1)
    valu,waves=m(c['SQ_INSTS_VALU']),m(c['SQ_WAVES'])
    us=m(dur[k])
    busy_us=valu*4/(1024*$CLK_GHZ*1e3)
    out[k]={"launches_averaged":len(half(c['SQ_WAVES'])),"waves_per_launch":waves,"valu_per_wave":valu/max(waves,1),"salu_per_wave":m(c['SQ_INSTS_SALU'])/max(waves,1),
            "lds_per_wave":m(c['SQ_INSTS_LDS'])/max(waves,1),"duration_us_under_pmc":us,"valu_issue_us":busy_us,"valu_issue_share":busy_us/us if us else None}
json.dump({"images_per_launch":2*$NP,"shader_clock_ghz":$CLK_GHZ,"note":"counters serialise kernels and lengthen them slightly; durations here are from the counter pass","kernels":out},open('gpurun_out/${NAME}_issue.json','w'),indent=1)
print(json.dumps(out,indent=1))
PY
