#!/bin/bash
# Repeats the poisoned-device-memory slice of the GPU suite (tests/test_poisoned_memory_gpu.py) N times and keeps the log of every failing run:
# a hunt for results that depend on uninitialised memory only now and then.   usage (on the GPU box): bash tools/poison_loop.sh [N]
N=${1:-40}
mkdir -p gpurun_out/poison
fail=0
for i in $(seq 1 $N); do
  PS_DEBUG_FILL=255 PS_BA_FILL=255 timeout 1200 python -m pytest -q -m gpu -x -k "not sweep and not poisoned" tests/test_opt_gpu.py tests/test_match_gpu.py tests/test_tracker_gpu.py tests/test_orb_gpu.py > gpurun_out/poison/run.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then fail=$((fail + 1)); cp gpurun_out/poison/run.log gpurun_out/poison/fail$i.log; echo "run $i rc=$rc: $(tail -1 gpurun_out/poison/run.log)"; fi
done
echo "poisoned slice: $fail failing runs of $N"
