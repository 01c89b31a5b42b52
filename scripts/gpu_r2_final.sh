#!/bin/bash
# round 2, final evidence run: whole GPU suite; the driver's bench command; kernel stats, HBM counters and SQ counters of the default
# workload; bench lines of the other workloads.  Everything lands in gpurun_out/<tag>/ and is copied to profiles/r2_* by hand.
set -u
TAG=${1:-r2_final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > "$OUT/smoke.log"; cat "$OUT/smoke.log"
timeout 1500 python -m pytest tests -m gpu -q --durations=10 --tb=short > "$OUT/pytest_gpu_full.log" 2>&1; tail -22 "$OUT/pytest_gpu_full.log" > "$OUT/pytest_gpu.log"; grep -n "^E  \|^FAILED\|passed\|failed" "$OUT/pytest_gpu_full.log" | head -20
SECONDS=0; ( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> "$OUT/bench.err" | tail -1 ) > "$OUT/bench_line.json"; echo "driver bench wall: ${SECONDS}s" | tee "$OUT/bench_wall.txt"; cut -c1-300 "$OUT/bench_line.json"; echo; tail -2 "$OUT/bench.err"
cd /tmp
B="--steps 6 --warmup 2 --preroll 40 --no-cpu-baseline --as-written 0 --check-channels 0"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o stats -- python "$GRAFT_REPO_ROOT/bench.py" $B > "$OUT/bench_prof_line.json" 2> "$OUT/prof.err"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && grep -E "k_oqpsk|k_coarse|Name" "$f" | cut -c1-200
KRE='k_oqpsk|k_coarse'
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc $c --output-format csv -d "$OUT/pmc_$c" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" $B > "$OUT/pmc_$c.log" 2>&1
  f=$(find "$OUT/pmc_$c" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/pmc_$c.csv"
done
cd "$GRAFT_REPO_ROOT"
bash scripts/pmc_sq.sh $TAG/sq $B > "$OUT/sq.log" 2>&1
for wl in msk burst_oqpsk burst_msk aerol aerol_burst aerol_c oqpsk8400; do
  extra=""; [ $wl = oqpsk8400 ] && extra="--as-written 0"
  ( timeout 600 python bench.py --workload $wl $extra 2> "$OUT/bench_$wl.err" | tail -1 ) > "$OUT/bench_line_$wl.json"; cut -c1-200 "$OUT/bench_line_$wl.json"; echo
done
find "$OUT" -name "*.csv" -size +6M -delete
du -sh "$OUT"
