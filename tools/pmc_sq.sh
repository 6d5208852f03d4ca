#!/bin/bash
# Issue-slot accounting per kernel of one bench step: one rocprofv3 PMC pass (8 SQ slots) over bench.py --steps 1.
# Prints, per kernel: wave quad-cycles split into active / parked (s_waitcnt, barrier) / issue-stalled, the share of
# active cycles spent in vector-ALU instructions, and matrix-pipe busy cycles per SIMD-cycle of the launch.
# usage (GPU box, repo root): tools/pmc_sq.sh [outdir]
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_sq}
mkdir -p "$OUT"
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d /tmp/pmc_sq -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs --no-other > "$OUT/bench.log" 2>&1
echo "rc=$?"
python - "$(find /tmp/pmc_sq -name '*counter_collection.csv' | head -1)" > "$OUT/summary.txt" <<'PY'
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
print("# per launch; wave cycles in quad-cycles summed over waves; mfma_busy in cycles summed over SIMDs")
print(f"{'kernel':58s} {'n':>4s} {'wave_qc':>11s} {'active':>7s} {'parked':>7s} {'stalled':>7s} {'valu/act':>8s} {'valu_insts':>11s} {'mfma_busy/wave_c':>16s}")
order = sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"])
for k in order[:16]:
    d, n = acc[k], len(disp[k])
    w = d["SQ_WAVE_CYCLES"] or 1.0
    print(f"{k[:58]:58s} {n:4d} {w / n:11.0f} {d['SQ_ACTIVE_INST_ANY'] / w:7.3f} {d['SQ_WAIT_ANY'] / w:7.3f} {d['SQ_WAIT_INST_ANY'] / w:7.3f} "
          f"{d['SQ_ACTIVE_INST_VALU'] / max(d['SQ_ACTIVE_INST_ANY'], 1):8.3f} {d['SQ_INSTS_VALU'] / n:11.0f} {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * w):16.3f}")
PY
cat "$OUT/summary.txt"
