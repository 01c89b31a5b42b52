#!/bin/bash
# round 2, call L: re-run the two fixed tests; counter calibration; the full default bench line; kernel stats + HBM counter passes of it;
# bench lines of the other workloads
set -u
TAG=${1:-r2l}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_scale.py::test_burst_oqpsk_4096_channels tests/test_jfastfir_vectors.py tests/test_gpu_parity.py::test_msk_live_set_settings tests/test_qt_adaptor.py -m gpu -q --tb=short > "$OUT/pytest_fixed.log" 2>&1; tail -3 "$OUT/pytest_fixed.log"; grep -n "^E  " "$OUT/pytest_fixed.log" | head
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/cal_$c" -o pmc -- "$GRAFT_REPO_ROOT/scripts/ubench/hbm_counters" > "$OUT/cal_$c.log" 2>&1
  f=$(find "$OUT/cal_$c" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/cal_$c.csv"
done
head -2 "$OUT/cal_FETCH_SIZE.log"
cd "$GRAFT_REPO_ROOT"
( timeout 900 python bench.py --steps 20 --warmup 5 2> "$OUT/bench.err" | tail -1 ) > "$OUT/bench_line.json"; cut -c1-400 "$OUT/bench_line.json"; tail -2 "$OUT/bench.err"
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
for wl in aerol_c msk burst_oqpsk; do
  ( timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 2> "$OUT/bench_$wl.err" | tail -1 ) > "$OUT/bench_line_$wl.json"; cut -c1-300 "$OUT/bench_line_$wl.json"; echo
done
find "$OUT" -name "*.csv" -size +6M -delete
du -sh "$OUT"
