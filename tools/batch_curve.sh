#!/bin/bash
# batch-size curve of the headline bench: one JSON line per batch size under gpurun_out/$1/
out=gpurun_out/${1:-curve}; mkdir -p $out
for b in ${BATCHES:-8 16 32 48 63 64 128}; do
  timeout 300 python bench.py --batch $b --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-configs --no-other > $out/b$b.json 2> $out/b$b.err
  python - $out/b$b.json $b <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print("B=%s %.1f crops/s %.2f ms | "%(sys.argv[2],d["value"],d["ms_per_step"])+" ".join("%s %.2f"%(n,v["ms_per_step"]) for n,v in k.items()))
except Exception as e: print("B=%s failed"%sys.argv[2], e)
PY
done
