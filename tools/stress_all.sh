#!/bin/bash
# The randomised GPU-vs-checker sweeps (tools/stress_*.py) over many seeds: a hunt for shapes on which the kernels and the CPU checker differ.
# usage (on the GPU box): bash tools/stress_all.sh [first seed] [seeds per tool]
S0=${1:-770001}; N=${2:-6}
fail=0
for i in $(seq 0 $((N - 1))); do
  seed=$((S0 + i))
  for job in "stress_orb.py $seed 120" "stress_cvorb_batch.py $seed 16" "stress_matchers.py $seed" "stress_opt.py $seed"; do
    out=$(timeout 900 python tools/$job 2>&1); rc=$?
    echo "$job rc=$rc: $(echo "$out" | tail -1 | cut -c1-160)"
    if [ $rc -ne 0 ]; then fail=$((fail + 1)); echo "$out" | tail -30; fi
  done
done
echo "stress sweeps: $fail failing jobs"
