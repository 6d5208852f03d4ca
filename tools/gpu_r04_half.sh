#!/bin/bash
# 256 x 128 plane-GEMM tiles below half a tile per slot: tests, then the ViT-L forward at 4 .. 32 crops with the tiles on / off, two rounds
python -m pytest tests/test_gpu_split.py -q -k "planes256 or vit" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python -m pytest tests/test_gpu_vit.py -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for r in 1 2; do for B in 4 8 12 16 24 32; do for h in 0 1; do
  echo -n "B=$B half=$h: "; GIGAPOSE_PLANES_HALF=$h python tools/probe_vit_loop.py $B 20 2>/dev/null | tail -1
done; done; done
