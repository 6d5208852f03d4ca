#!/bin/bash
# thin plane arithmetic everywhere (GEMM epilogues 6 / 7, LayerNorm planes, attention P split, convolution epilogues): tests, then the
# bench step alternately against a library built from the previous commit (GIGAPOSE_LIB)
BASE=$PWD/gigapose_amd/libgigapose_hip_base.so
python -m pytest tests/test_gpu_split.py tests/test_gpu_vit.py tests/test_gpu_conv.py tests/test_gpu_e2e.py tests/test_gpu_guards.py tests/test_gpu_lnfold.py -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for r in 1 2 3; do for lib in base new; do
  if [ $lib = base ]; then export GIGAPOSE_LIB=$BASE; else unset GIGAPOSE_LIB; fi
  python bench.py --no-cpu-baseline --no-configs --no-other --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['kernels'].items() if k in ('gemm_split','layernorm','attention','conv')})
"; done; done
