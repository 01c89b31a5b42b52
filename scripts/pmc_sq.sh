#!/bin/bash
# SQ issue / wait counters of the two headline kernels (kernel-trace + --pmc only).  usage: scripts/pmc_sq.sh <tag> [bench flags]
set -u
TAG=${1:-sq}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
KRE='k_oqpsk|k_msk|k_coarse|k_burst|k_hilbert|k_trident|k_pre8400|k_viterbi|k_aerol'
timeout 200 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/a" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline > "$OUT/a.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --output-format csv -d "$OUT/b" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline > "$OUT/b.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# a step's sample-loop work is one full launch plus a one-sample launch: average the full launches only (values >= 10 % of the largest)
def full(v):
    m = max(v)
    w = [x for x in v if x >= 0.1 * m] if m > 0 else v
    return sum(w) / len(w), len(w)
summ = {k: {c: full(v)[0] for c, v in d.items()} | {"full_launches": max(full(v)[1] for v in d.values()), "launches": max(len(v) for v in d.values())}
        | ({"SQ_INSTS_VALU_sum": sum(d["SQ_INSTS_VALU"])} if "SQ_INSTS_VALU" in d else {}) for k, d in res.items()}
import os
summ["tag"] = os.environ.get("SQ_TAG", "untagged")
# steps of work the profiled command issued (warm-up + timed [+ re-written steps]): the Aero-L lines price a whole step against VALU issue
if os.environ.get("STEPS_TOTAL"): summ["steps_total"] = float(os.environ["STEPS_TOTAL"])
summ["channels_per_gpu"] = int(os.environ.get("CHANNELS", "65536"))
json.dump(summ, open(out + "/sq_summary.json", "w"), indent=1)
print(json.dumps(summ, indent=1))
PY
find "$OUT" -name "*.csv" -size +4M -delete
