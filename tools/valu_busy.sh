#!/bin/bash
# developer tool: how busy the vector ALUs / the LDS are per kernel of the step: one lockstep group of 512 sequences (kernels one after the other).
# A plain run first: the counter passes read the rendered sequences from the cache it leaves.
A="--no-cpu --no-secondary --no-alone --groups 1 --sequences 512 --steps 3 --warmup 2"
python3 bench.py $A > /dev/null 2>&1
R=$PWD; export TMPDIR=/tmp
pass() {
  cd /tmp; rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/$1 -o pmc -- python3 $R/bench.py $A > $R/gpurun_out/$1.log 2>&1; cd $R
}
pass vb1 "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
pass vb2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
python3 - <<'PY'
import csv,glob,collections
tab=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for d in ('vb1','vb2'):
    f=glob.glob('gpurun_out/%s/*counter_collection.csv'%d)
    if not f: print('no csv for',d); continue
    seen=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:34]
        tab[k][d+':'+r['Counter_Name']]+=float(r['Counter_Value'])
out={}
print('%-36s %9s %9s %9s %8s %8s %8s %8s' % ('kernel (all launches of the run)', 'VALU busy', 'SALU busy', 'waves/SIMD', 'wait', 'LDS busy', 'conflict', 'VALU/wave'))
for k,v in sorted(tab.items(), key=lambda kv: -kv[1].get('vb1:GRBM_GUI_ACTIVE',0)):
    gui=v.get('vb1:GRBM_GUI_ACTIVE',0)/8.0            # cycles of the launch(es): the counter is summed over the 8 XCDs
    if gui<=0: continue
    simd=gui*1024.0
    valu=4*v.get('vb1:SQ_ACTIVE_INST_VALU',0)/simd
    occ=4*v.get('vb1:SQ_WAVE_CYCLES',0)/simd
    wait=v.get('vb1:SQ_WAIT_INST_ANY',0)/max(v.get('vb1:SQ_WAVE_CYCLES',1),1)
    gui2=v.get('vb2:GRBM_GUI_ACTIVE',0)/8.0
    lds=v.get('vb2:SQ_LDS_IDX_ACTIVE',0)/(gui2*256.0) if gui2 else 0
    conf=v.get('vb2:SQ_LDS_BANK_CONFLICT',0)/max(v.get('vb2:SQ_LDS_IDX_ACTIVE',1),1)
    sca=4*v.get('vb2:SQ_ACTIVE_INST_SCA',0)/(gui2*1024.0) if gui2 else 0      # one scalar unit per CU, one instruction per cycle
    print('%-36s %8.2f %9.2f %9.1f %8.2f %8.2f %8.2f %9.0f' % (k, valu, sca, occ, wait, lds, conf, v.get('vb1:SQ_INSTS_VALU',0)/max(v.get('vb1:SQ_WAVES',1),1)))
    base=k.split('<')[0].split('(')[0].strip()
    if base.startswith('at::') or base.startswith('__amd') or 'elementwise' in base: continue
    e=out.setdefault(base, {'cycles':0.0,'valu':0.0,'salu':0.0,'lds':0.0,'waves_per_simd':0.0})
    # template instances of one kernel (orb_level_fused<true/false>, orb_quadtree<...>, cvb_select<...>): busy fractions weighted by their cycles
    for key,val in (('valu',valu),('salu',sca),('lds',lds),('waves_per_simd',occ)): e[key]=(e[key]*e['cycles']+val*gui)/(e['cycles']+gui)
    e['cycles']+=gui
import json
tot=sum(e['cycles'] for e in out.values()) or 1.0
for e in out.values():
    e['share_of_kernel_cycles']=round(e['cycles']/tot,5); e['cycles']=round(e['cycles'])
    for key in ('valu','salu','lds','waves_per_simd'): e[key]=round(e[key],4)
json.dump({'what':'per kernel of the step (one lockstep group of 512 sequences, kernels one after the other): fraction of the kernel\'s cycles its vector ALUs (4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE x 1024 SIMDs)), the CUs\' scalar units (SQ_ACTIVE_INST_SCA) and the LDS (SQ_LDS_IDX_ACTIVE / (cycles x 256 CUs)) were busy; two separate --pmc passes, tools/valu_busy.sh',
           'sequences':512,'kernels':out}, open('gpurun_out/unit_busy.json','w'), indent=1)
PY
