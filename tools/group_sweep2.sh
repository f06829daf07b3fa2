#!/bin/bash
for cfg in "1536 3" "2048 4" "2560 5" "1536 3" "2048 4"; do
  set -- $cfg
  python3 bench.py --no-cpu --no-secondary --no-alone --sequences $1 --groups $2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sequences=$1 groups=$2 %.0f ms/step %.2f' % (d['value'], d['ms_per_step']))"
done
