#!/bin/bash
# A/B of the thin plane epilogues 6 / 7 (a second library built with -DGP_THIN_EPI): bit-identity tests, then the bench step alternately
T=$PWD/gigapose_amd/libgigapose_hip_thin.so
GIGAPOSE_LIB=$T python -m pytest tests/test_gpu_split.py tests/test_gpu_vit.py -q -k "planes256 or vit_large" 2>&1 | tail -2
for r in 1 2 3; do for lib in base thin; do
  if [ $lib = thin ]; then export GIGAPOSE_LIB=$T; else unset GIGAPOSE_LIB; fi
  python bench.py --no-cpu-baseline --no-configs --no-other --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['kernels'].items() if k in ('gemm_split','layernorm')})
"; done; done
