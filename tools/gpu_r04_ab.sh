#!/bin/bash
# A/B of the ViT-L forward: folded LayerNorm in place (1), ping-pong (2), unfolded (0); two rounds against box drift
for r in 1 2; do for f in 1 2 0; do echo -n "fold=$f: "; GIGAPOSE_LN_FOLD=$f python tools/probe_vit_loop.py 64 10 2>/dev/null | tail -1; done; done
