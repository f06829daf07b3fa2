#!/bin/bash
# Regenerates EXACTLY the committed profiles/<round>_* files on an MI355X box (run through gpurun from the repository root):
#
#   gpurun --timeout 2400 -- 'tools/reproduce_profiles.sh r05'      ->  gpurun_out/profiles/r05_*   (copy into profiles/ and commit)
#
#   <round>_bench.json, <round>_bench_full.json      the driver's command (python3 bench.py --gpus 1 --steps 20 --warmup 5): line + full result (run LAST)
#   <round>_unit_busy.json, _valu_busy.txt           vector ALU / scalar unit / LDS busy fractions per kernel (tools/valu_busy.sh)
#   <round>_s1_chain.txt, _s1_chain_kernel_stats.csv one SLOT.MODE-4 sequence per handle, one frame in flight (BASELINE configs[4] per GPU)
#   <round>_shim_path.txt, _shim_kernel_stats.csv, _shim_memory_copy_stats.csv   examples/stereo_kitti.cpp on the shim classes, host images
#   <round>_headline_runs.txt                        the headline loop alone in five fresh processes (spread between runs on one box)
#   <round>_group_sweep.txt                          lockstep groups x sequences per GPU
#   <round>_bench_kernel_stats[_timed].csv           rocprofv3 --kernel-trace --stats of the headline loop (3 groups) / its timed launches only
#   <round>_bench_kernel_stats_1group[_timed].csv    the same loop with --groups 1 (the pass `roofline` is measured in)
#   <round>_overlap_table.txt                        what the overlap of the groups does to every kernel (from the two traces above)
#   <round>_traffic.json, _valu_issue.json, _pmc_table.txt   HBM bytes / instruction issue per kernel of the step: three separate --pmc passes
#                                                    (request size classes TCC_EA0_RDREQ_{32,64,128}B, TCC_EA0_WRREQ(_64B); SQ_INSTS_*), never with a trace domain
#   <round>_<leg>_kernel_stats.csv, _legs_pmc.json, _legs_table.txt   the legs outside the step: ba, ba1, bf, cfse3, pose (tools/*_quick.py)
#
# Under rocprofv3 the program itself follows `--` (python3 <script>): the profiler initialises the GPU before the program starts, so
# nothing in between may exec.  The counter passes read their sequences from the cache the first plain run leaves (bench.py PS_SEQ_CACHE).
set -u
ROUND=${1:-r06}
R=$PWD
OUT=$R/gpurun_out/profiles
W=$R/gpurun_out/$ROUND
mkdir -p $OUT $W
export TMPDIR=/tmp
STEPS=20; WARM=5
HEAD="--no-cpu --no-secondary --no-alone --steps $STEPS --warmup $WARM"

# ---- 2. spread of the headline over fresh processes; group sweep ----
{ echo "# python3 bench.py $HEAD, five fresh processes on one box: tracked frames/s, ms per step"
  for i in 1 2 3 4 5; do python3 bench.py $HEAD 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))"; done
} > $OUT/${ROUND}_headline_runs.txt
{ echo "# python3 bench.py --no-cpu --no-secondary --no-alone --sequences S --groups G: tracked frames/s, ms per step"
  for C in "256 1" "512 1" "768 1" "512 2" "768 3" "1024 4" "1536 3" "1536 2" "2304 3" "3072 3"; do set -- $C
    python3 bench.py --no-cpu --no-secondary --no-alone --sequences $1 --groups $2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sequences=$1 groups=$2', round(d['value']), 'ms/step', round(d['ms_per_step'],2))"
  done
} > $OUT/${ROUND}_group_sweep.txt

# ---- 3. kernel trace + stats of the headline loop, three groups and one ----
for V in "" "_1group"; do
  EXTRA=""; [ -n "$V" ] && EXTRA="--groups 1"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt$V -o orb -- python3 $R/bench.py $HEAD $EXTRA > $W/kt$V.log 2>&1 || true
  cd $R
  S=$(find $W/kt$V -name "*kernel_stats.csv" | head -1); T=$(find $W/kt$V -name "*kernel_trace.csv" | head -1)
  [ -n "$S" ] && cp $S $OUT/${ROUND}_bench_kernel_stats$V.csv
  [ -n "$T" ] && python3 tools/trace_stats.py $T $STEPS $WARM > $OUT/${ROUND}_bench_kernel_stats${V}_timed.csv
done
python3 tools/overlap_table.py $W/kt 3 $STEPS $WARM > $OUT/${ROUND}_overlap_table.txt 2>&1 || true

# ---- 4. counters of the step: separate passes, kernel trace only beside them ----
pmc() {   # <tag> <counters...>
  TAG=$1; shift
  cd /tmp
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/${ROUND}_$TAG -o pmc -- python3 $R/bench.py $HEAD --steps 3 --warmup 2 > $R/gpurun_out/${ROUND}_$TAG.log 2>&1 || true
  cd $R
}
pmc rd TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B
pmc wr TCC_EA0_WRREQ TCC_EA0_WRREQ_64B
pmc sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
python3 tools/pmc_step_table.py $ROUND 3 2 > $OUT/${ROUND}_pmc_table.txt
for f in traffic valu_issue; do [ -f gpurun_out/${ROUND}_$f.json ] && cp gpurun_out/${ROUND}_$f.json $OUT/; done

# ---- 5. the legs outside the step ----
script_of() { case $1 in ba) echo "tools/ba_quick.py 8";; ba1) echo "tools/ba_quick.py 1";; bf) echo "tools/bf_quick.py";; cfse3) echo "tools/cfse3_quick.py 4";; pose) echo "tools/pose_quick.py 64";; esac; }
LEGS="ba ba1 bf cfse3 pose"
mkdir -p $R/gpurun_out/${ROUND}_legs
for leg in $LEGS; do
  S=$(script_of $leg)
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${ROUND}_legs/${leg}_kt -o kt -- python3 $R/$S > $R/gpurun_out/${ROUND}_legs/${leg}_kt.log 2>&1 || true
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B --output-format csv -d $R/gpurun_out/${ROUND}_legs/${leg}_rd -o pmc -- python3 $R/$S > $R/gpurun_out/${ROUND}_legs/${leg}_rd.log 2>&1 || true
  rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B --output-format csv -d $R/gpurun_out/${ROUND}_legs/${leg}_wr -o pmc -- python3 $R/$S > $R/gpurun_out/${ROUND}_legs/${leg}_wr.log 2>&1 || true
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/${ROUND}_legs/${leg}_sq -o pmc -- python3 $R/$S > $R/gpurun_out/${ROUND}_legs/${leg}_sq.log 2>&1 || true
  cd $R
done
python3 tools/legs_table.py ${ROUND}_legs $LEGS > $OUT/${ROUND}_legs_table.txt
for leg in $LEGS; do [ -f gpurun_out/${ROUND}_legs_${leg}_kernel_stats.csv ] && cp gpurun_out/${ROUND}_legs_${leg}_kernel_stats.csv $OUT/${ROUND}_${leg}_kernel_stats.csv; done
[ -f gpurun_out/${ROUND}_legs_legs_pmc.json ] && cp gpurun_out/${ROUND}_legs_legs_pmc.json $OUT/${ROUND}_legs_pmc.json

# ---- 6. unit-busy fractions per kernel of the step (two more counter passes, tools/valu_busy.sh) ----
tools/valu_busy.sh > $OUT/${ROUND}_valu_busy.txt 2>&1 || true
[ -f gpurun_out/unit_busy.json ] && cp gpurun_out/unit_busy.json $OUT/${ROUND}_unit_busy.json

# ---- 7. BASELINE configs[4]'s per-GPU share: ONE SLOT.MODE-4 sequence, one frame in flight (tools/s1_chain_bench.py) ----
{ python3 tools/s1_chain_bench.py 60 1 1 2>/dev/null | grep -v amdgpu.ids; PS_TRK_OVERLAP=0 python3 tools/s1_chain_bench.py 60 1 1 2>/dev/null | grep "S=1" | sed 's/$/   (PS_TRK_OVERLAP=0: one stream)/'
  python3 tools/s1_chain_bench.py 60 0 1 2>/dev/null | grep "S=1"; } > $OUT/${ROUND}_s1_chain.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $W/s1 -o s1 -- python3 $R/tools/s1_chain_bench.py 60 1 1 > $W/s1.log 2>&1 || true
cd $R
S=$(find $W/s1 -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/${ROUND}_s1_chain_kernel_stats.csv

# ---- 8. the drop-in boundary: examples/stereo_kitti.cpp on the shim classes, host images (kernels AND copies traced) ----
python3 - <<PY
import sys
sys.path.insert(0, "$R")
import bench
from pointslot_amd import sequence
sequence.write_pgm("/tmp/ps_cfg5_seq", bench._config5_sequence(0, 154))
PY
./build/stereo_kitti /tmp/ps_cfg5_seq | tail -6 > $OUT/${ROUND}_shim_path.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $W/shim -o shim -- $R/build/stereo_kitti /tmp/ps_cfg5_seq > $W/shim.log 2>&1 || true
cd $R
S=$(find $W/shim -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/${ROUND}_shim_kernel_stats.csv
S=$(find $W/shim -name "*memory_copy_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/${ROUND}_shim_memory_copy_stats.csv

# ---- 9. full-size long-chain parity record (tests/test_track_device_gpu.py writes gpurun_out/config5_chain_parity.json) ----
python3 -m pytest tests/test_track_device_gpu.py -m gpu -q -k config5_length > $W/config5_parity.log 2>&1 || true
[ -f gpurun_out/config5_chain_parity.json ] && cp gpurun_out/config5_chain_parity.json $OUT/${ROUND}_config5_chain_parity_full.json

# ---- 10. the driver's command, LAST: with this box's counter tables in profiles/ (bench.py reads its traffic / unit-busy figures there) ----
for f in traffic valu_issue unit_busy legs_pmc; do [ -f $OUT/${ROUND}_$f.json ] && cp $OUT/${ROUND}_$f.json profiles/; done
python3 bench.py --gpus 1 --steps $STEPS --warmup $WARM > $W/bench_line.txt 2> $W/bench_stderr.txt
tail -1 $W/bench_line.txt > $OUT/${ROUND}_bench.json
cp bench_full.json $OUT/${ROUND}_bench_full.json

ls -la $OUT
