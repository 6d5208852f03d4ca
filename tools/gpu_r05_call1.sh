#!/bin/bash
# round 5, GPU call 1: the full GPU suite (new: plane scales, accumulated drop-in flow, config-5 e2e) + the default bench line
set -x
OUT=gpurun_out/r05_call1
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s --durations=25 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
echo skip bench


