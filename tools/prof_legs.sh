#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats and three separate counter passes (--pmc only with --kernel-trace)
# of the legs that have no place in the headline step's profile:
#   ba     tools/ba_quick.py 8     object BA, config 4 (8 objects x 50 KF x 300 points)            a16 / metric_ba
#   ba1    tools/ba_quick.py 1     one object alone
#   bf     tools/bf_quick.py       SearchByBruceMatching 1 / 8 / 64 x 1000 x 1000                  a10
#   cfse3  tools/cfse3_quick.py 4  CFSE3ObjStateOptimization 64 frames x 4 objects x 150 points    a15
#   pose   tools/pose_quick.py 64  PoseOptimization 64 frames x 2000 edges                         a14
# Usage: tools/prof_legs.sh <name> [legs...]   ->  gpurun_out/<name>/<leg>_{kt,rd,wr,sq}/ and, through tools/legs_table.py,
# gpurun_out/<name>_<leg>_kernel_stats.csv + gpurun_out/<name>_legs_pmc.json
NAME=${1:-legs}; shift || true
LEGS=${@:-ba ba1 bf cfse3 pose}
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$NAME
script_of() {
  case $1 in
    ba) echo "tools/ba_quick.py 8";; ba1) echo "tools/ba_quick.py 1";; bf) echo "tools/bf_quick.py";;
    cfse3) echo "tools/cfse3_quick.py 4";; pose) echo "tools/pose_quick.py 64";;
  esac
}
for leg in $LEGS; do
  S=$(script_of $leg)
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$NAME/${leg}_kt -o kt -- python3 $R/$S > $R/gpurun_out/$NAME/${leg}_kt.log 2>&1 || true
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B --output-format csv -d $R/gpurun_out/$NAME/${leg}_rd -o pmc -- python3 $R/$S > $R/gpurun_out/$NAME/${leg}_rd.log 2>&1 || true
  rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B --output-format csv -d $R/gpurun_out/$NAME/${leg}_wr -o pmc -- python3 $R/$S > $R/gpurun_out/$NAME/${leg}_wr.log 2>&1 || true
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/$NAME/${leg}_sq -o pmc -- python3 $R/$S > $R/gpurun_out/$NAME/${leg}_sq.log 2>&1 || true
  cd $R
done
python3 tools/legs_table.py $NAME $LEGS
