#!/bin/bash
# 256 x 128 tiles, second form (128 < tiles <= 208: stream-K over all slots): tests, then the ViT-L forward with
# half = 0 (off) / 128 (parallel form only) / 1 (both, default)
python -m pytest tests/test_gpu_split.py -q -k "planes256 or vit" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python -m pytest tests/test_gpu_vit.py tests/test_gpu_guards.py -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for r in 1 2; do for B in 12 16 20 24 40 48; do for h in 0 128 1; do
  echo -n "B=$B half=$h: "; GIGAPOSE_PLANES_HALF=$h python tools/probe_vit_loop.py $B 20 2>/dev/null | tail -1
done; done; done
