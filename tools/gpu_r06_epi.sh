set -u
export TMPDIR=/tmp
O=gpurun_out/r06_epi
mkdir -p $O
echo "new:";  python tools/feature_hash.py 2>&1 | tail -3
echo "base:"; GIGAPOSE_LIB=gigapose_amd/libbase.so python tools/feature_hash.py 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_split.py tests/test_gpu_pose_ist.py tests/test_gpu_e2e.py tests/test_gpu_guards.py -x -q 2>&1 | tail -3
for r in 1 2 3; do
for v in new base; do
L=gigapose_amd/libgigapose_hip.so; [ $v = base ] && L=gigapose_amd/libbase.so
GIGAPOSE_LIB=$L python bench.py --steps 20 --no-cpu-baseline --no-configs --no-other > $O/bench_${v}_$r.json 2> $O/bench_${v}_$r.err
python - <<PY
import json
d=json.load(open("$O/bench_${v}_$r.json")); k=d["roofline"]["kernels"]
print("$v", d["value"], d["ms_per_step"], "other", k["other"]["ms_per_step"], "gemm_split", k["gemm_split"]["ms_per_step"], d["roofline"]["executed_tflops"], d["roofline"]["sustained_mfma_only_tflops"], {b: d["batch_curve"][b]["value"] for b in ("b8","b16","b32")} if "batch_curve" in d else "")
PY
done
done
