set -u
export TMPDIR=/tmp
OUT=gpurun_out/pmc_attn
mkdir -p $OUT
run() { name=$1; shift; timeout 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pa/$name -o p -- python tools/probe_attn_split.py --plain > $OUT/$name.log 2>&1; echo "pass $name rc=$?"; f=$(find /tmp/pa/$name -name "*counter_collection.csv" | head -1); python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:50]
    if "attention_split" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
disp = len({r["Dispatch_Id"] for r in rows if "attention_split" in r["Kernel_Name"]})
for k, d in acc.items():
    for c, v in sorted(d.items()): print(f"{c:34s} {v/disp:16.0f} per launch ({disp} launches)")
PY
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU
