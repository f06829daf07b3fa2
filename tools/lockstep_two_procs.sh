#!/bin/bash
# developer tool: N lockstep driver processes on one GPU at the same time (is the limit inside one process?)  usage: tools/lockstep_two_procs.sh <procs> <seqs> <groups>
NP=${1:-2}; S=${2:-128}; G=${3:-2}
python3 - <<PY
import sys; sys.path.insert(0, ".")
from pointslot_amd import sequence
for k in range(4): sequence.write_pgm("/tmp/ps_ls/%04d" % k, sequence.generate(n_frames=14, seed=30 + k, step=0.05 + 0.01 * k))
PY
ARGS=""; for i in $(seq 0 $((S - 1))); do ARGS="$ARGS /tmp/ps_ls/000$((i % 4))"; done
for p in $(seq 1 $NP); do build/stereo_kitti_batch --groups $G $ARGS > /tmp/ps_ls_out_$p.txt 2>&1 & done
wait
for p in $(seq 1 $NP); do tail -1 /tmp/ps_ls_out_$p.txt | cut -c1-210; done
