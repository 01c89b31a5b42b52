#!/bin/bash
# HBM counter passes of the MSK workload with the pair kernel
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2x/msk; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
B="--workload msk --steps 6 --warmup 2 --no-cpu-baseline"
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --kernel-include-regex 'k_msk|k_coarse' --pmc $c --output-format csv -d "$OUT/pmc_$c" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" $B > "$OUT/pmc_$c.log" 2>&1
  f=$(find "$OUT/pmc_$c" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/pmc_$c.csv"
  rm -rf "$OUT/pmc_$c"
done
ls -la "$OUT"
