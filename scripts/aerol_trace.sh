#!/bin/bash
# per-launch durations of the Aero-L pipeline kernels (rocprofv3 kernel trace of a short bench run)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-aerol_trace}
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --workload aerol --steps 4 --warmup 2 --no-cpu-baseline > "$OUT/line.json" 2> "$OUT/err.txt"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
for k, v in sorted(d.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    if not any(t in k for t in ("aerol", "viterbi")): continue
    v.sort()
    print(k, "n=%d total=%.2f ms" % (len(v), sum(x[1] for x in v)), "last launches ms:", " ".join("%.3f" % x[1] for x in v[-9:]))
PY
