#!/bin/bash
# rocprofv3 kernel statistics of a short bench run; usage: tools/prof_stats.sh <tag> [env assignments...]
# writes gpurun_out/prof_<tag>/ and prints the top kernels (per-kernel average duration, calls, share)
tag=$1; shift
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
env "$@" rocprofv3 --kernel-trace --stats -d $out -o run -- python $root/bench.py --no-cpu-baseline --no-configs --no-other --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err
cd $root
f=$(find $out -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n[:110]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':112s} {'calls':>6s} {'avg us':>9s} {'total ms':>9s} {'%':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f"{short(r['Name']):112s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:9.2f} {100*float(r['TotalDurationNs'])/tot:6.1f}")
PY
