#!/bin/bash
# in-pipeline cost of the parts of the folded producer epilogue: ViT-L forward under rocprofv3 with the probe masks of
# gp_gemm_planes256_set_dp (bits 2..5); per-kernel averages printed here (the rocpd databases are too large to travel)
export TMPDIR=/tmp
root=$PWD
for m in ${MASKS:-1 9 17 33 off}; do
  out=/tmp/vitprof_$m; rm -rf $out; mkdir -p $out
  if [ $m = off ]; then envs="GIGAPOSE_LN_FOLD=0"; else envs="GIGAPOSE_LN_FOLD=1 GIGAPOSE_PLANES_DP=$m"; fi
  echo "== mask $m ($envs)"
  (cd /tmp && env $envs rocprofv3 --kernel-trace -d $out -o run -- python $root/tools/probe_vit_loop.py 64 6 2>/dev/null | tail -1)
  python tools/rocprof_summary.py $(find $out -name "*.db" | head -1) 7 | cut -c1-150
done
