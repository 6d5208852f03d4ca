set -u
export TMPDIR=/tmp
O=gpurun_out/final_prof
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-other > $O/bench_under_prof.log 2>&1
grep '^{"metric' $O/bench_under_prof.log > $O/bench_under_prof.json
DB=$(find /tmp/kt -name "*.db" | head -1)
python tools/rocprof_summary.py $DB 50 > $O/kernel_stats.txt 2>&1
python tools/step_gaps.py $DB 8 4 > $O/step_gaps.txt 2>&1
bash tools/pmc_bench.sh $O/pmc_bench > $O/pmc_bench.log 2>&1
bash tools/pmc_sq.sh $O/pmc_sq > $O/pmc_sq.log 2>&1
head -3 $O/step_gaps.txt
