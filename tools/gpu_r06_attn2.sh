# round 6: attention with chunked staging -- tests, A/B of the staging order, bench
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_attn2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_vit.py tests/test_gpu_plane_scales.py tests/test_gpu_checkpoint.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
python tools/probe_attn_split.py > $O/probe_attn.txt 2>&1; head -8 $O/probe_attn.txt
python bench.py --no-cpu-baseline --no-other --no-configs > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step']); print({k:(v['ms_per_step'], v['avg_launch_us']) for k,v in d['roofline']['kernels'].items()})"
