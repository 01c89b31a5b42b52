#!/bin/bash
# burst OQPSK workload with the overlap-save Hilbert kernel: bench line + rocprofv3 kernel stats
set -u
TAG=${1:-r2r}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
( timeout 600 python bench.py --workload burst_oqpsk 2> "$OUT/bench.err" | tail -1 ) > "$OUT/bench_line.json"; cut -c1-300 "$OUT/bench_line.json"; echo
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o stats -- python "$GRAFT_REPO_ROOT/bench.py" --workload burst_oqpsk --no-cpu-baseline > "$OUT/bench_prof_line.json" 2> "$OUT/prof.err"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "^\"Name|k_" "$f" | cut -c1-200 > "$OUT/kernel_stats.csv"; cat "$OUT/kernel_stats.csv"
find "$OUT" -name "*.csv" -size +6M -delete
