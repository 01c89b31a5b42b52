#!/bin/bash
# Samples the GPU's engine clock and power (rocm-smi) every 0.25 s while a bench workload runs: what clock do the fp64-bound kernels get?
# usage: scripts/diag/clock_during_bench.sh <workload> [bench args...]   -> gpurun_out/clock_<workload>.txt
WL=${1:-oqpsk}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/clock_$WL.txt
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > "$OUT" &
MON=$!
python bench.py --workload $WL --no-cpu-baseline --check-channels 0 --as-written 0 "$@" 2>/dev/null | tail -1 | cut -c1-200
kill $MON
sort "$OUT" | uniq -c | sort -rn | head -12
