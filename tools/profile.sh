#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the headline loop of the default bench (bench.py --no-cpu --no-secondary), CSV output under
# gpurun_out/<name>/.  Usage: tools/profile.sh <name> [bench args...]
set -e
NAME=${1:-prof}; shift || true
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$NAME -o orb -- python3 $R/bench.py --no-cpu --no-secondary "$@" > $R/gpurun_out/$NAME.log 2>&1 || true
cd $R
# the --stats averages include the warm-up launches (the first launch of a kernel pays the code-object load); the same table over
# the timed launches only, from the kernel trace of this run (bench.py defaults: 20 timed steps after 3 warm-up steps)
python3 tools/trace_stats.py gpurun_out/$NAME/orb_kernel_trace.csv ${PS_PROF_STEPS:-20} ${PS_PROF_WARMUP:-3} > gpurun_out/$NAME/orb_kernel_stats_timed.csv
find gpurun_out/$NAME -name "*stats*.csv" | head
