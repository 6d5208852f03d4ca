#!/bin/bash
# round 5: where does the B = 16 step spend its GEMM time?  kernel trace of bench.py --batch 16 / 24 / 48 + the tests of the touched kernels
export TMPDIR=/tmp
O=gpurun_out/r05_b16; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_vit.py -m gpu -q -k "parallel or half_width or every_batch or planes256" > $O/pytest.log 2>&1
tail -2 $O/pytest.log
for B in 16 48; do
  rm -rf /tmp/kt$B
  timeout 300 rocprofv3 --kernel-trace -d /tmp/kt$B -o kt -- python bench.py --batch $B --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-other > $O/bench$B.log 2>&1
  DB=$(find /tmp/kt$B -name "*.db" | head -1)
  python tools/rocprof_summary.py $DB 14 > $O/kernels_b$B.txt 2>&1
  head -10 $O/kernels_b$B.txt | cut -c1-170
done
BATCHES="8 12 16 24 32 48 63" STEPS=10 bash tools/batch_curve.sh r05_curve
