#!/bin/bash
# round 2, call A: the whole GPU test suite (incl. the BASELINE-size banks and the un-gated C-channel banks), then first numbers
# for the two f4 halves with rocprofv3 kernel stats.
set -u
TAG=${1:-r2a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -60 ) > "$OUT/pytest_gpu.log"
tail -5 "$OUT/pytest_gpu.log"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof8400" -o stats -- python "$GRAFT_REPO_ROOT/scripts/bench_8400.py" 16384 8 > "$OUT/bench_8400.json" 2> "$OUT/bench_8400.err"
cat "$OUT/bench_8400.json"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/profaerolc" -o stats -- python "$GRAFT_REPO_ROOT/scripts/bench_aerol_c.py" 4096 8 > "$OUT/bench_aerolc.json" 2> "$OUT/bench_aerolc.err"
cat "$OUT/bench_aerolc.json"
cd "$GRAFT_REPO_ROOT"
for d in prof8400 profaerolc; do f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${d}_kernel_stats.csv" && head -12 "$f"; done
find "$OUT" -name "*.csv" -size +4M -delete
du -sh "$OUT"
