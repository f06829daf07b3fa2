#!/bin/bash
# samples the GPU's clock and power while the headline loop runs
( for i in $(seq 120); do echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | tr '\n' ' ' | tr -s ' ')"; sleep 0.5; done ) > gpurun_out/clk_samples.txt &
SP=$!
python3 bench.py --steps 60 --warmup 5 --no-secondary --no-cpu --no-alone 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%8.0f frames/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"
kill $SP 2>/dev/null
grep -c . gpurun_out/clk_samples.txt
awk 'NR%4==0' gpurun_out/clk_samples.txt | cut -c1-200 | tail -25
