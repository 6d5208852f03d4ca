#!/bin/bash
# half-width tiles with three k-steps of loads in flight: tests, then the ViT-L forward against the library of the previous commit (one step in flight)
BASE=$PWD/gigapose_amd/libgigapose_hip_head.so
python -m pytest tests/test_gpu_split.py -q -k "planes256 or vit" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python -m pytest tests/test_gpu_vit.py -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for r in 1 2; do for B in 4 8 12 16 32; do for lib in base new; do
  if [ $lib = base ]; then export GIGAPOSE_LIB=$BASE; else unset GIGAPOSE_LIB; fi
  echo -n "B=$B $lib: "; python tools/probe_vit_loop.py $B 20 2>/dev/null | tail -1
done; done; done
