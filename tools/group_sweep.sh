# developer tool: headline loop over lockstep group counts / sequence counts, optionally with more HIP hardware queues
# usage: tools/group_sweep.sh "<queues> <sequences> <groups>" ...
for C in "$@"; do
  set -- $C
  GPU_MAX_HW_QUEUES=$1 python bench.py --no-cpu --no-secondary --no-alone --sequences $2 --groups $3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('queues=$1 sequences=$2 groups=$3', round(d['value']), 'ms/step', round(d['ms_per_step'],2))"
done
