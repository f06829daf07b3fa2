#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the headline loop of the default bench (bench.py --no-cpu --no-secondary: two lockstep
# groups on two streams) and of the same loop with ONE group (--groups 1: the kernels never overlap - the pass bench.py's `roofline` is measured in), CSV
# output under gpurun_out/<name>/ and gpurun_out/<name>_1group/.  Usage: tools/profile.sh <name> [bench args...]
set -e
NAME=${1:-prof}; shift || true
R=$PWD
export TMPDIR=/tmp
for V in "" "_1group"; do
  EXTRA=""; [ -n "$V" ] && EXTRA="--groups 1"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$NAME$V -o orb -- python3 $R/bench.py --no-cpu --no-secondary --no-alone $EXTRA "$@" > $R/gpurun_out/$NAME$V.log 2>&1 || true
  cd $R
  # the --stats averages include the warm-up launches (the first launch of a kernel pays its code-object load); the same table over
  # the timed launches only, from the kernel trace of this run (bench.py defaults: 20 timed steps after 3 warm-up steps; the default run
  # would add a single-group pass behind the timed region: --no-alone)
  python3 tools/trace_stats.py gpurun_out/$NAME$V/orb_kernel_trace.csv ${PS_PROF_STEPS:-20} ${PS_PROF_WARMUP:-3} > gpurun_out/$NAME$V/orb_kernel_stats_timed.csv
done
find gpurun_out/$NAME gpurun_out/${NAME}_1group -name "*stats*.csv" | head
