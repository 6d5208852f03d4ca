#!/bin/bash
# convolution tile prologue (one round trip instead of four) and tail (residual requested ahead): tests, then the bench step
# alternately against a library built from the previous commit (GIGAPOSE_LIB), then per-kernel times of both
BASE=$PWD/gigapose_amd/libgigapose_hip_head.so
python -m pytest tests/test_gpu_conv.py tests/test_gpu_pose_ist.py tests/test_gpu_e2e.py tests/test_gpu_guards.py -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for r in 1 2 3; do for lib in base new; do
  if [ $lib = base ]; then export GIGAPOSE_LIB=$BASE; else unset GIGAPOSE_LIB; fi
  python bench.py --no-cpu-baseline --no-configs --no-other --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['kernels'].items() if k in ('gemm_split','layernorm','attention','conv')})
"; done; done
cd /tmp && export TMPDIR=/tmp
for lib in base new; do
  if [ $lib = base ]; then export GIGAPOSE_LIB=$BASE; else unset GIGAPOSE_LIB; fi
  rocprofv3 --kernel-trace -d /tmp/kt_$lib -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-other > /dev/null 2>&1
  echo "== $lib"; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt_$lib -name "*.db" | head -1) 40 | grep -E "conv_" | cut -c1-150
done
