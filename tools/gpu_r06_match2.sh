set -u
export TMPDIR=/tmp
O=gpurun_out/r06_match2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_matcher.py tests/test_gpu_parity_big.py -x -q 2>&1 | tail -3
python tools/probe_match_fixed.py > $O/timeline.txt 2>&1; tail -30 $O/timeline.txt
python bench.py --steps 20 --no-cpu-baseline --no-configs --no-other > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json")); k=d["roofline"]["kernels"]
print(d["value"], d["ms_per_step"], {n:(v["ms_per_step"]) for n,v in k.items()})
print({n:v for n,v in d["roofline"].items() if n.startswith("sustained") or n.startswith("executed") or n=="matrix_ceiling_ubench"})
PY
