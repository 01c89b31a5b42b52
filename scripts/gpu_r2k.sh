#!/bin/bash
# round 2, call K: whole GPU suite; f4 bench lines (8400 bps demodulator, C-channel Aero-L) with rocprofv3 kernel stats
set -u
TAG=${1:-r2k}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q --durations=8 --tb=short > "$OUT/pytest_gpu_full.log" 2>&1; tail -15 "$OUT/pytest_gpu_full.log" > "$OUT/pytest_gpu.log"; grep -n "^E  \|^FAILED\|passed\|failed" "$OUT/pytest_gpu_full.log" | head -30
cd /tmp
for wl in oqpsk8400 aerol_c; do
  extra=""; [ $wl = oqpsk8400 ] && extra="--preroll 12 --check-channels 8 --as-written 0"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$wl" -o stats -- python "$GRAFT_REPO_ROOT/bench.py" --workload $wl --steps 8 --warmup 2 $extra > "$OUT/bench_$wl.json" 2> "$OUT/bench_$wl.err"
  tail -1 "$OUT/bench_$wl.json" | cut -c1-1500
  f=$(find "$OUT/prof_$wl" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${wl}_kernel_stats.csv" && grep -E "k_|Name" "$f" | cut -c1-160 | head -12
done
find "$OUT" -name "*.csv" -size +4M -delete
du -sh "$OUT"
