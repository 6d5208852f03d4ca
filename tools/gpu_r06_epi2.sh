set -u
export TMPDIR=/tmp
O=gpurun_out/r06_epi
mkdir -p $O
for r in 1 2; do
for v in new base; do
L=gigapose_amd/libgigapose_hip.so; [ $v = base ] && L=gigapose_amd/libbase.so
GIGAPOSE_LIB=$L python bench.py --steps 10 --no-cpu-baseline --no-other > $O/benchfull_${v}_$r.json 2> $O/benchfull_${v}_$r.err
python - <<PY
import json
d=json.load(open("$O/benchfull_${v}_$r.json"))
bc=d.get("batch_curve",{})
print("$v", d["value"], {b: (bc[b]["value"], bc[b]["ms_per_step"]) for b in bc if b.startswith("b") and isinstance(bc[b], dict) and "value" in bc[b]}, {c: d["other_configs"][c]["value"] for c in d.get("other_configs",{})})
PY
done
done
