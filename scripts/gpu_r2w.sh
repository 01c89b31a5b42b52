#!/bin/bash
# HBM counter passes (FETCH_SIZE, WRITE_SIZE; kernel-trace + --pmc only) of the 8400 bps and MSK workloads
set -u
TAG=${1:-r2w}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
for wl in oqpsk8400 msk; do
  B="--workload $wl --steps 6 --warmup 2 --no-cpu-baseline"
  [ $wl = oqpsk8400 ] && B="$B --as-written 0 --check-channels 0 --preroll 40"
  mkdir -p "$OUT/$wl"
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --kernel-include-regex 'k_oqpsk|k_msk|k_coarse|k_pre8400' --pmc $c --output-format csv -d "$OUT/$wl/pmc_$c" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" $B > "$OUT/$wl/pmc_$c.log" 2>&1
    f=$(find "$OUT/$wl/pmc_$c" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/$wl/pmc_$c.csv"
  done
  rm -rf "$OUT/$wl/pmc_WRITE_SIZE" "$OUT/$wl/pmc_FETCH_SIZE"
  ls -la "$OUT/$wl"
done
find "$OUT" -name "*.csv" -size +8M -delete
