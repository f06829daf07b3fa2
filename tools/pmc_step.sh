#!/bin/bash
# HBM traffic and instruction issue of every kernel of the headline step from PMC counters, run on the GPU box:
#   tools/pmc_step.sh <name>     ->  gpurun_out/<name>_traffic.json, gpurun_out/<name>_valu_issue.json
# Three separate counter passes (--pmc only with --kernel-trace) over `bench.py --no-cpu --no-secondary --distinct 2` (two distinct
# sequences are generated in-process: the counter tool has the GPU open before python starts, so no process pool may be spawned):
#   reads   TCC_EA0_RDREQ and its size classes _32B / _64B / _128B: bytes = 32 n32 + 64 n64 + 128 n128 (checked: n32 + n64 + n128 = n).
#           This replaces FETCH_SIZE (= RDREQ x 64 B on gfx950, half of the bytes when the requests are 128 B wide, exact when they
#           are 64 B wide - MI355X_MICROARCH.md, HBM) by the request sizes themselves: no calibration factor per access pattern.
#   writes  TCC_EA0_WRREQ and TCC_EA0_WRREQ_64B: bytes = 64 n64 + 32 (n - n64)
#   issue   SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_WAVES
NAME=${1:-pmc}; STEPS=${2:-3}; WARM=${3:-2}
R=$PWD
export TMPDIR=/tmp
run() {   # <tag> <counters>
  cd /tmp
  rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/${NAME}_$1 -o pmc -- python3 $R/bench.py --no-cpu --no-secondary --no-alone --distinct 2 --steps $STEPS --warmup $WARM > $R/gpurun_out/${NAME}_$1.log 2>&1 || true
  cd $R
}
run rd "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B"
run wr "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"
run sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
python3 tools/pmc_step_table.py $NAME $STEPS $WARM
