#!/bin/bash
# IST backbone on a second stream (bench.py --overlap) below 64 crops, where neither chain fills the chip
for B in 8 16 32; do for o in "" "--overlap"; do
  echo -n "B=$B ${o:-single}: "; python bench.py --batch $B --steps 20 $o --no-cpu-baseline --no-configs --no-other 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])"
done; done
