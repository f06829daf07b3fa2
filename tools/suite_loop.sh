#!/bin/bash
# Repeats the whole GPU suite N times (fresh process each) and keeps the log of every failing run - a hunt for tests that fail now and then.
# usage (on the GPU box): bash tools/suite_loop.sh [N]
N=${1:-3}
mkdir -p gpurun_out/suite
fail=0
for i in $(seq 1 $N); do
  s=$(date +%s)
  timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/suite/run.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(( $(date +%s) - s )) s: $(tail -1 gpurun_out/suite/run.log)"
  if [ $rc -ne 0 ]; then fail=$((fail + 1)); cp gpurun_out/suite/run.log gpurun_out/suite/fail$i.log; fi
done
echo "GPU suite: $fail failing runs of $N"
