set -u
export TMPDIR=/tmp
O=gpurun_out/r06_gelu
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_split.py tests/test_gpu_vit.py tests/test_gpu_plane_scales.py tests/test_gpu_guards.py -x -q -s 2>&1 | grep -E "GELU epilogues|passed|failed|Error|assert" | tail -8
for r in 1 2 3; do
for v in new base; do
L=gigapose_amd/libgigapose_hip.so; [ $v = base ] && L=gigapose_amd/libbase.so
GIGAPOSE_LIB=$L python bench.py --steps 20 --no-cpu-baseline --no-configs --no-other > $O/bench_${v}_$r.json 2> $O/bench_${v}_$r.err
python - <<PY
import json
d=json.load(open("$O/bench_${v}_$r.json")); k=d["roofline"]["kernels"]
print("$v", d["value"], d["ms_per_step"], "gemm_split", k["gemm_split"]["ms_per_step"], d["roofline"]["executed_tflops"], d["roofline"]["sustained_mfma_only_tflops"])
PY
done
done
PYTHONPATH=. python tools/probe_planes_timeline.py 2>&1 | grep -E "^fc1" | cut -c1-420
timeout 900 python -m pytest tests/test_gpu_parity_big.py tests/test_gpu_e2e.py -x -q -s 2>&1 | grep -E "same_all|hyp|passed|failed|Error|assert" | tail -30
