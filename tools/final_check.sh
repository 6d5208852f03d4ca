#!/bin/bash
# Last check of a round on the GPU box: the whole -m gpu suite, smoke(), the default bench line.  Outputs under gpurun_out/final_check/.
set -u
O=gpurun_out/final_check
mkdir -p $O
( time python -m pytest tests -m gpu -q -s ) > $O/pytest_gpu.log 2>&1   # -s: every test prints its measured differences
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/bench.log 2>&1
grep '^{"metric' $O/bench.log > $O/bench.json
grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
tail -1 $O/smoke.log | cut -c1-200
cut -c1-220 $O/bench.json
