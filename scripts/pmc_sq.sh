#!/bin/bash
# SQ issue / wait counters of the two headline kernels (kernel-trace + --pmc only).  usage: scripts/pmc_sq.sh <tag> [bench flags]
set -u
TAG=${1:-sq}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
KRE='k_oqpsk|k_msk|k_coarse'
timeout 200 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/a" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline --steps 3 --warmup 2 > "$OUT/a.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --output-format csv -d "$OUT/b" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline --steps 3 --warmup 2 > "$OUT/b.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": max(len(v) for v in d.values())} for k, d in res.items()}
json.dump(summ, open(out + "/sq_summary.json", "w"), indent=1)
print(json.dumps(summ, indent=1))
PY
find "$OUT" -name "*.csv" -size +4M -delete
