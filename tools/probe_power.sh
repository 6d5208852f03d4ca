#!/bin/bash
# Is the step power-limited?  Samples socket power, the power cap and the shader clock (rocm-smi) once a second while bench.py runs a
# long timed region, in both numerics.  Usage (GPU box): bash tools/probe_power.sh > gpurun_out/power.txt
rocm-smi --showmaxpower --showpowercap 2>/dev/null | grep -iE "power|cap" | head -6
for mode in split chain; do
  echo "== numerics=$mode"
  python bench.py --numerics $mode --steps 300 --warmup 3 --no-cpu-baseline --no-configs --no-other > /tmp/bench_$mode.json 2>/dev/null &
  BP=$!
  for i in $(seq 1 90); do
    if ! kill -0 $BP 2>/dev/null; then break; fi
    echo "t=$i $(rocm-smi -P -c -u --json 2>/dev/null | python -c "
import json,sys
try:
    d=json.load(sys.stdin)['card0']
    print({k:v for k,v in d.items() if any(s in k.lower() for s in ('power','sclk','use'))})
except Exception as e: print('?', e)
")"
    sleep 1
  done
  wait $BP
  python -c "
import json
d=json.load(open('/tmp/bench_$mode.json'))
print('bench', d['value'], d['ms_per_step'])
"
done
