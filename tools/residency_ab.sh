#!/bin/bash
# experiment (r05): fewer workgroups of the issue-bound extraction kernels per CU (extra dynamic LDS) so that the other lockstep groups'
# kernels find wave slots / LDS beside them; the headline loop alone per setting, two runs each
run() {
  label=$1; shift
  for rep in 1 2; do
    env "$@" python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu --no-alone 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-44s %8.0f frames/s  %.3f ms/step' % ('$label', d['value'], d['ms_per_step']))"
  done
}
run "default (FAST 8 / CU, levels 7 / CU)" PS_DUMMY=1
run "FAST 6 per CU" PS_FAST_LDS_PAD=6000
run "FAST 5 per CU" PS_FAST_LDS_PAD=12000
run "FAST 4 per CU" PS_FAST_LDS_PAD=20000
run "levels 5 per CU" PS_LV_LDS_PAD=21000
run "levels 4 per CU" PS_LV_LDS_PAD=29000
run "FAST 6 + levels 5" PS_FAST_LDS_PAD=6000 PS_LV_LDS_PAD=21000
run "FAST 5 + levels 4 + describe 4" PS_FAST_LDS_PAD=12000 PS_LV_LDS_PAD=29000 PS_DESC_LDS_PAD=8000
