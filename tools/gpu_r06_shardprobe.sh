# round 6: kernel lists of one 64-crop step, unsharded vs sharded (forced one-rank RCCL), and the all-padding step
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_shardprobe
mkdir -p $O
for mode in plain sharded; do
  MODE=$mode timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_$mode -o kt -- python tools/probe_sharded_flow.py 10 > $O/$mode.log 2>&1
  DB=$(find /tmp/kt_$mode -name "*.db" | head -1)
  python tools/rocprof_summary.py $DB 60 > $O/kernels_$mode.txt 2>&1
done
MODE=sharded timeout 300 python tools/probe_sharded_flow.py 10 0 > $O/sharded_live0.log 2>&1
MODE=sharded timeout 300 python tools/probe_sharded_flow.py 10 40 > $O/sharded_live40.log 2>&1
MODE=sharded timeout 300 python tools/probe_sharded_flow.py 10 > $O/sharded_noprof.log 2>&1
MODE=plain timeout 300 python tools/probe_sharded_flow.py 10 > $O/plain_noprof.log 2>&1
grep -h "ms per" $O/*.log
