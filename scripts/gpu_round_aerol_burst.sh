#!/bin/bash
# Burst-mode Aero-L (row f2) on the GPU box: its parity tests, bench line, rocprofv3 kernel stats.  usage: scripts/gpu_round_aerol_burst.sh <tag>
set -u
TAG=${1:-aerol_burst}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
( timeout 600 python -m pytest tests/test_gpu_aerol_burst.py -q 2>&1 | tail -5 ) > "$OUT/pytest_gpu.log"; tail -2 "$OUT/pytest_gpu.log"
( timeout 600 python bench.py --workload aerol_burst 2> "$OUT/bench.err" | tail -1 ) > "$OUT/bench_line.json"; cat "$OUT/bench_line.json"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o stats -- python "$GRAFT_REPO_ROOT/bench.py" --workload aerol_burst --no-cpu-baseline > "$OUT/bench_prof_line.json" 2> "$OUT/prof.err"
cd "$GRAFT_REPO_ROOT"
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
with open(out + "/kernel_stats.csv", "w") as fo:
    w = csv.writer(fo); w.writerow(["kernel", "calls", "total_ms", "avg_us", "max_us", "pct"])
    for r in rows[:12]:
        w.writerow([r["Name"].split("(")[0][:60], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 3), round(float(r["AverageNs"]) / 1e3, 2), round(float(r["MaxNs"]) / 1e3, 2), r["Percentage"]])
print(open(out + "/kernel_stats.csv").read())
PY
find "$OUT" -name "*.csv" -size +8M -delete; du -sh "$OUT"
