#!/bin/bash
# PMC passes (one rocprofv3 run per counter group; FETCH_SIZE and WRITE_SIZE never share a pass: MI355X guide,
# "rocprofv3 PMC slots") over the small per-kernel workload tools/probe_pmc.py.
# usage: tools/pmc_passes.sh <outdir>      (run on the GPU box from the repo root)
set -u
OUT=${1:-gpurun_out/pmc}
export TMPDIR=/tmp
mkdir -p "$OUT"
run() { # name, counters...
  local name=$1; shift
  timeout 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- python tools/probe_pmc.py > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum
python tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
find "$OUT" -name "*.db" -delete
