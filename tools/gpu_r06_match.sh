# round 6: the matcher's tile order inside an XCD chunk (bands of 2 / 4 / 8 / 16 crops x all templates): launch time inside the bench
# step and L2-miss traffic per launch (FETCH_SIZE pass), one variant library per band size (tools/build_variant.sh)
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_match
mkdir -p $O
for v in gigapose_hip match_b2 match_b4 match_b16; do
  export GIGAPOSE_LIB=$PWD/gigapose_amd/lib$v.so
  python bench.py --steps 10 --no-cpu-baseline --no-configs --no-other > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
d=json.load(open("$O/bench_$v.json")); k=d["roofline"]["kernels"]
print("$v", d["value"], d["ms_per_step"], "match_split", k["match_split"]["ms_per_step"], "gemm", k["gemm_split"]["ms_per_step"])
PY
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pm_$v -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs --no-other > $O/pmc_$v.log 2>&1
  python - "$(find /tmp/pm_$v -name '*counter_collection.csv' | head -1)" "$v" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "match_tiles_split" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
disp={}
for r in rows: disp[r["Dispatch_Id"]]=disp.get(r["Dispatch_Id"],0.0)+float(r["Counter_Value"])
big=[v for v in disp.values() if v>1e5]
print(sys.argv[2], "match_tiles_split FETCH_SIZE x 2 KiB per launch:", [round(v*2*1024/1e9,3) for v in big], "GB")
PY
done
unset GIGAPOSE_LIB
python tools/probe_match_fixed.py > $O/timeline.txt 2>&1; tail -14 $O/timeline.txt
