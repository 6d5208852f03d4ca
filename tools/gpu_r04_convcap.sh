#!/bin/bash
# parallel split of the convolutions: at least n channel blocks per slot of a split tile (GIGAPOSE_CONV_HALO = 1 + 16 n); bench step at 8 / 16 / 32 crops
for r in 1 2; do for B in 8 16 32; do for n in 1 2 4; do
  echo -n "B=$B min_cb=$n: "; GIGAPOSE_CONV_HALO=$((1 + 16 * n)) python bench.py --batch $B --steps 10 --no-cpu-baseline --no-configs --no-other 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['kernels'].items() if k in ('conv','gemm_split')})"
done; done; done
