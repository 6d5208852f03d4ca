#!/bin/bash
# round 5: LayerNorm in 16-token blocks below 256 blocks -- tests + A/B of the step at 8 / 16 / 24 crops (GIGAPOSE_LN_REG=2 = the 32-token blocks)
export TMPDIR=/tmp
O=gpurun_out/r05_ln16; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vit.py tests/test_gpu_split.py -m gpu -q -k "every_batch or peaked or vit_large or layerwise or aenet" 2>&1 | grep -E "passed|failed|^FAILED" | tail -3
for rep in 1 2; do
for mode in 2 1; do
  for b in 8 16 24; do
    GIGAPOSE_LN_REG=$mode timeout 300 python bench.py --batch $b --steps 12 --warmup 3 --no-cpu-baseline --no-configs --no-other > $O/m${mode}_b$b.json 2>/dev/null
    python - $O/m${mode}_b$b.json $mode $b <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("LN_REG=%s B=%s %.1f crops/s %.3f ms | layernorm %.3f ms (%s us/launch)"%(sys.argv[2],sys.argv[3],d["value"],d["ms_per_step"],k["layernorm"]["ms_per_step"],k["layernorm"]["avg_launch_us"]))
PY
  done
done
done
