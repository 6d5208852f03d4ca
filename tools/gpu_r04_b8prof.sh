#!/bin/bash
# per-kernel times of the whole step at 8 crops (kernel trace of bench.py --batch 8)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/kt8 -o kt -- python $GRAFT_REPO_ROOT/bench.py --batch ${1:-8} --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-other > /tmp/b8.log 2>&1
grep '^{"metric' /tmp/b8.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt8 -name "*.db" | head -1) 34 | cut -c1-150
python $GRAFT_REPO_ROOT/tools/step_gaps.py $(find /tmp/kt8 -name "*.db" | head -1) 8 4 | head -12
