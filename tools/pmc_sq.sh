#!/bin/bash
# Shader-sequencer counters of the ORB kernels (developer tool, run on the GPU box): three --pmc passes over tools/orb_quick_bench.py,
# per-kernel means of the later launches, written to gpurun_out/<name>_sq.json.  Usage: tools/pmc_sq.sh <name> [npairs] [script]
NAME=${1:-sq}; NP=${2:-64}; SCRIPT=${3:-tools/orb_quick_bench.py}
R=$PWD
export TMPDIR=/tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM"
P2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
P3="SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VSKIPPED SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  cd /tmp
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/${NAME}_sq$i -o pmc -- python3 $R/$SCRIPT $NP > $R/gpurun_out/${NAME}_sq$i.log 2>&1 || true
  cd $R
done
python3 - <<PY
import csv,glob,collections,json
def short(k): return k.replace('(anonymous namespace)::','').split('(')[0].replace('void ','')
out=collections.defaultdict(dict)
for i in (1,2,3):
    f=glob.glob('gpurun_out/${NAME}_sq%d/*counter_collection.csv' % i)
    if not f: continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in agg.items():
        for n,v in c.items():
            v=v[len(v)//2:]
            out[k][n]=sum(v)/len(v)
    t=glob.glob('gpurun_out/${NAME}_sq%d/*kernel_trace.csv' % i)
    dur=collections.defaultdict(list)
    for r in csv.DictReader(open(t[0])):
        dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k,v in dur.items():
        v=v[len(v)//2:]
        out[k]['duration_us_pass%d' % i]=sum(v)/len(v)
res={}
for k,c in out.items():
    if not (k.startswith('orb_') or k.startswith('st_') or k.startswith('pj_') or k.startswith('pose_') or k.startswith('trk_')): continue
    w=c.get('SQ_WAVES',0) or 1
    c['per_wave']={n[9:].lower(): c[n]/w for n in c if n.startswith('SQ_INSTS_')}
    res[k]=c
json.dump({"images_per_launch":2*$NP,"kernels":res},open('gpurun_out/${NAME}_sq.json','w'),indent=1)
for k,c in res.items():
    print(k, json.dumps({a:(round(b,1) if not isinstance(b,dict) else {x:round(y,1) for x,y in b.items()}) for a,b in c.items()}))
PY
