# round 6: attention with fewer vector instructions -- tests, launch time, instruction counters, bench
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_attn
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_vit.py tests/test_gpu_plane_scales.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
bash tools/pmc_attn.sh > $O/pmc_attn.txt 2>&1; grep -E "SQ_INSTS_VALU|SQ_VALU_MFMA|SQ_ACTIVE_INST_VALU|SQ_ACTIVE_INST_ANY|GRBM" $O/pmc_attn.txt
python bench.py --no-cpu-baseline --no-other --no-configs > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step']); print({k:(v['ms_per_step'], v['avg_launch_us']) for k,v in d['roofline']['kernels'].items()})"
