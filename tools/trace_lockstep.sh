#!/bin/bash
# developer tool: rocprofv3 kernel statistics of the lockstep tracker (64 sequences, one group); run on the GPU box
R=$PWD
export TMPDIR=/tmp
python3 - <<PY
import os, sys
sys.path.insert(0, "$R")
from pointslot_amd import sequence
for k in range(4):
    sequence.write_pgm("/tmp/ps_ls/%04d" % k, sequence.generate(n_frames=12, seed=30 + k, step=0.05 + 0.01 * k))
PY
ARGS=""
for i in $(seq 0 $((${PS_SEQS:-64} - 1))); do ARGS="$ARGS /tmp/ps_ls/000$((i % 4))"; done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lockstep -o ls -- $R/build/stereo_kitti_batch --groups ${PS_GROUPS:-1} $ARGS > $R/gpurun_out/lockstep.log 2>&1
cd $R
tail -2 gpurun_out/lockstep.log
f=$(find gpurun_out/prof_lockstep -name "*kernel_stats.csv" | head -1)
cut -c1-160 $f | head -24
