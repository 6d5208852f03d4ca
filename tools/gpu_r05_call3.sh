#!/bin/bash
# round 5, GPU call 3: fixed / new tests, the fc2-in-parts measurements (stage errors, end-to-end parity counts, step time)
set -x
OUT=gpurun_out/r05_call3
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dropin_flow.py tests/test_gpu_e2e.py tests/test_gpu_guards.py tests/test_gpu_plane_scales.py tests/test_gpu_matcher.py \
    tests/test_gpu_pose_ist.py tests/test_gpu_split.py tests/test_gpu_conv.py -m gpu -q -s > $OUT/pytest_sel.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_sel.log
grep -n "passed\|failed\|^FAILED" $OUT/pytest_sel.log | tail -12
timeout 300 python tools/probe_stage_errors.py > $OUT/stage_errors.txt 2>&1
tail -22 $OUT/stage_errors.txt
for P in 0 2 4; do
  GIGAPOSE_FC2_PARK=$P timeout 600 python -m pytest tests/test_gpu_parity_big.py -m gpu -q -s -k "benchmark_size and split" > $OUT/e2e_park$P.log 2>&1
  echo "park $P rc $?"
  grep -h "vs the reference in float64\|ViT-L unit-norm features" $OUT/e2e_park$P.log | cut -c1-700
  GIGAPOSE_FC2_PARK=$P timeout 300 python bench.py --no-cpu-baseline --no-configs --no-other --steps 20 > $OUT/bench_park$P.json 2> $OUT/bench_park$P.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_park$P.json").read().strip().splitlines()[-1])
    k = d["roofline"]["kernels"]
    print("park $P:", d["value"], "crops/s", d["ms_per_step"], "ms; gemm_split", k["gemm_split"]["ms_per_step"], "ms/step")
except Exception as e:
    print("park $P: parse error", e)
PY
done
