#!/bin/bash
# k-steps per slot floor of the parallel split (GIGAPOSE_PLANES_PAR = n) with the half-width tiles: ViT-L forward at small batches
for r in 1 2; do for B in 4 8 12 16; do for n in 16 8 4; do
  echo -n "B=$B min_steps=$n: "; GIGAPOSE_PLANES_PAR=$n python tools/probe_vit_loop.py $B 20 2>/dev/null | tail -1
done; done; done
