# developer tool: headline loop with pose_lm variants (build_exp/libps_<tag>.so), value and pose stage times
for V in "" "$@"; do
  if [ -z "$V" ]; then L=""; else L="build_exp/libps_$V.so"; fi
  PS_LIB_PATH=$L python bench.py --no-cpu --no-secondary --no-alone 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['stage_ms']
print('variant=${V:-base}', round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'pose', s['pose_optimization'], 'cfse3', s['object_cfse3'])"
done
