#!/bin/bash
# developer tool: the headline loop alone (no secondary legs, no CPU baseline) once per library variant, same box, back to back.
# usage: tools/ab_bench.sh [-r repeats] <tag|default> ...    (build_exp/libps_<tag>.so from tools/build_variant.sh)
R=2
if [ "$1" = "-r" ]; then R=$2; shift 2; fi
for rep in $(seq $R); do
for v in "$@"; do
  if [ "$v" = default ]; then unset PS_LIB_PATH; else export PS_LIB_PATH=build_exp/libps_$v.so; fi
  python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu $AB_ARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
ka=d.get('kernels_alone',{}).get('stage_ms_per_512_sequences',{})
print('%-10s value %8.0f  ms/step %.3f | alone/512: orb %.3f objfeat %.3f pose %.3f cfse3 %.3f stereo %.3f sbp %.3f glue %.3f oglue %.3f bf %.3f' % ('$v', d['value'], d['ms_per_step'], ka.get('orb_extract',0), ka.get('object_features',0), ka.get('pose_optimization',0), ka.get('object_cfse3',0), ka.get('stereo_match',0), ka.get('search_by_projection',0), ka.get('track_glue',0), ka.get('object_glue',0), ka.get('object_bruteforce',0)))"
done
done
