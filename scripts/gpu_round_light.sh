#!/bin/bash
# Bench line (with the CPU baseline) + rocprofv3 kernel stats of one workload, no test suite, no PMC.  usage: scripts/gpu_round_light.sh <tag> [bench flags]
set -u
TAG=${1:-light}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
( timeout 600 python bench.py "$@" 2> "$OUT/bench.err" | tail -1 ) > "$OUT/bench_line.json"; cat "$OUT/bench_line.json"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o stats -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline --steps 8 --warmup 4 > "$OUT/bench_prof_line.json" 2> "$OUT/prof.err"
cd "$GRAFT_REPO_ROOT"
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
with open(out + "/kernel_stats.csv", "w") as fo:
    w = csv.writer(fo); w.writerow(["kernel", "calls", "total_ms", "avg_us", "max_us", "pct"])
    for r in rows[:10]:
        w.writerow([r["Name"].split("(")[0][:60], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 3), round(float(r["AverageNs"]) / 1e3, 2), round(float(r["MaxNs"]) / 1e3, 2), r["Percentage"]])
print(open(out + "/kernel_stats.csv").read())
PY
find "$OUT" -name "*.csv" -size +8M -delete
