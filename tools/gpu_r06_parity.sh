set -u
export TMPDIR=/tmp
O=gpurun_out/r06_parity
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_plane_scales.py tests/test_gpu_split.py -x -q 2>&1 | tail -2
timeout 2400 python tools/probe_parity_attribution.py "$@" > $O/attribution.txt 2> $O/attribution.err; tail -3 $O/attribution.err; cat $O/attribution.txt
