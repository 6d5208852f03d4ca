#!/bin/bash
# LayerNorm planes kernel with gamma | beta through LDS: tests, then the ViT-L forward at 8 / 16 / 64 crops against the previous commit's library
BASE=$PWD/gigapose_amd/libgigapose_hip_head.so
python -m pytest tests/test_gpu_vit.py tests/test_gpu_lnfold.py -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for r in 1 2 3; do for B in 8 16 64; do for lib in base new; do
  if [ $lib = base ]; then export GIGAPOSE_LIB=$BASE; else unset GIGAPOSE_LIB; fi
  echo -n "B=$B $lib: "; python tools/probe_vit_loop.py $B 20 2>/dev/null | tail -1
done; done; done
