#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the default bench, CSV output under
# gpurun_out/<name>/.  Usage: tools/profile.sh <name> [bench args...]
set -e
NAME=${1:-prof}; shift || true
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$NAME -o orb -- python3 $R/bench.py --no-cpu "$@" > $R/gpurun_out/$NAME.log 2>&1 || true
cd $R
find gpurun_out/$NAME -name "*stats*.csv" | head
