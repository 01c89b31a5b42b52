#!/bin/bash
# after the 8400 bps pair kernel: whole GPU suite, smoke, the driver's bench command, the 8400 bps workload with kernel stats
set -u
TAG=${1:-r2u}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 ) | tee "$OUT/smoke.log"
timeout 1500 python -m pytest tests -m gpu -q --durations=5 --tb=short > "$OUT/pytest_gpu_full.log" 2>&1; tail -8 "$OUT/pytest_gpu_full.log" | tee "$OUT/pytest_gpu.log"
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> "$OUT/bench.err" | tail -1 ) > "$OUT/bench_line.json"; cut -c1-200 "$OUT/bench_line.json"; echo
( timeout 600 python bench.py --workload oqpsk8400 --as-written 0 2> "$OUT/bench_8400.err" | tail -1 ) > "$OUT/bench_line_oqpsk8400.json"; cut -c1-200 "$OUT/bench_line_oqpsk8400.json"; echo
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o stats -- python "$GRAFT_REPO_ROOT/bench.py" --workload oqpsk8400 --steps 6 --warmup 2 --no-cpu-baseline --as-written 0 --check-channels 0 > "$OUT/bench_prof_line.json" 2> "$OUT/prof.err"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "^\"Name|k_" "$f" | cut -c1-200 > "$OUT/kernel_stats_8400.csv"; cat "$OUT/kernel_stats_8400.csv"
find "$OUT" -name "*.csv" -size +6M -delete
