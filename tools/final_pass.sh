set -u
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
python bench.py > $O/bench.log 2>&1
grep '^{"metric' $O/bench.log > $O/bench.json
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs > $O/bench_under_prof.log 2>&1
grep '^{"metric' $O/bench_under_prof.log > $O/bench_under_prof.json
DB=$(find /tmp/kt -name "*.db" | head -1)
python tools/rocprof_summary.py $DB 50 > $O/kernel_stats.txt 2>&1
python tools/step_gaps.py $DB 8 4 > $O/step_gaps.txt 2>&1
bash tools/pmc_bench.sh $O/pmc_bench > $O/pmc_bench.log 2>&1
bash tools/pmc_sq.sh $O/pmc_sq > $O/pmc_sq.log 2>&1
python tools/probe_planes_timeline.py > $O/planes_timeline.txt 2>&1
python tools/probe_conv_timeline.py > $O/conv_timeline.txt 2>&1
python tools/probe_match_fixed.py > $O/match_fixed.txt 2>&1
python tools/probe_attn_split.py > $O/attn_split.txt 2>&1
tail -3 $O/pytest_gpu.log | head -5
cut -c1-200 $O/bench.json
