#!/bin/bash
# round 4, GPU call: folded LayerNorm -- unit tests, the regression tests of the files it touched, an A/B of the ViT forward and of the bench step
set -x
export TMPDIR=/tmp
O=gpurun_out/r04_call4
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_lnfold.py -x -q -s > $O/lnfold.log 2>&1; echo "lnfold rc $?"; tail -25 $O/lnfold.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_matcher.py tests/test_gpu_split.py tests/test_gpu_vit.py tests/test_gpu_guards.py -x -q > $O/regress.log 2>&1; echo "regress rc $?"; tail -8 $O/regress.log | cut -c1-300
for f in 1 0; do GIGAPOSE_LN_FOLD=$f timeout 300 python bench.py --no-cpu-baseline --no-configs --no-other --steps 10 > $O/bench_fold$f.json 2> $O/bench_fold$f.err; echo "bench fold=$f rc $?"; python - <<PY
import json
d=json.load(open("$O/bench_fold$f.json"))
print("fold=$f", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], {k:(v["ms_per_step"], v["launches_per_step"]) for k,v in d["roofline"]["kernels"].items()})
PY
done
