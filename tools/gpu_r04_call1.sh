#!/bin/bash
# round 4, GPU call 1: the whole GPU suite on the phase-1 tree, the bench line with the new extras, the per-stage error probe
set -x
export TMPDIR=/tmp
O=gpurun_out/r04_call1
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee $O/pytest_rc.txt
tail -5 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee $O/bench_rc.txt
tail -c 1500 $O/bench.json
timeout 300 python tools/probe_stage_errors.py > $O/stage_errors.txt 2>&1; echo "probe rc $?"
cat $O/stage_errors.txt | tail -20
