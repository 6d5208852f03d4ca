#!/bin/bash
# HBM traffic of the bench step per kernel family: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE never share a
# pass) over one single-stream bench step; writes <out>/traffic.json (consumed by bench.py for roofline.traffic).
set -u
OUT=${1:-gpurun_out/pmc_bench}
export TMPDIR=/tmp
mkdir -p "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs ${PMC_BENCH_FLAGS:-} > "$OUT/$c.log" 2>&1
  echo "pass $c rc=$?"
done
python tools/pmc_summary.py "$OUT" --json "$OUT/traffic.json" > "$OUT/summary.txt" 2>&1
tail -30 "$OUT/summary.txt"
find "$OUT" -name "*.db" -delete
