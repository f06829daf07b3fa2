#!/bin/bash
# experiment: lockstep groups on CU partitions instead of all groups on all CUs; the headline loop alone per setting
# (r06: the knob is compiled in only with -DPS_DEV_CU_PARTITION: tools/build_all_variant.sh cupart -DPS_DEV_CU_PARTITION, then PS_LIB_PATH=build_exp/libps_cupart.so)
run() {  # label, env...
  label=$1; shift
  for rep in 1 2; do
    env "$@" python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu --no-alone $ARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-34s %8.0f frames/s  %.3f ms/step' % ('$label', d['value'], d['ms_per_step']))"
  done
}
ARGS=""
run "3 groups, all CUs (headline)" PS_DUMMY=1
run "3 groups, disjoint thirds" PS_CU_PARTITION=3
run "3 groups, two thirds each" PS_CU_PARTITION=3 PS_CU_SHARE=2
run "3 groups, 6 ranges, 3 each" PS_CU_PARTITION=6 PS_CU_SHARE=3
ARGS="--groups 2 --sequences 512"
run "2 groups x 256, all CUs" PS_DUMMY=1
run "2 groups x 256, disjoint halves" PS_CU_PARTITION=2
ARGS="--groups 4 --sequences 1024"
run "4 groups x 256, all CUs" PS_DUMMY=1
run "4 groups x 256, disjoint quarters" PS_CU_PARTITION=4
run "4 groups x 256, halves" PS_CU_PARTITION=4 PS_CU_SHARE=2
