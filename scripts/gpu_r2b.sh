#!/bin/bash
# round 2, call B: GPU test suite (no -x), the re-worked default bench (per-channel symbol-clock phases, pre-roll, oracle check), the
# same bench with symbol-synchronous channels for comparison, SQ counters of the default.
set -u
TAG=${1:-r2b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q --durations=8 --tb=short > "$OUT/pytest_gpu_full.log" 2>&1; tail -40 "$OUT/pytest_gpu_full.log" > "$OUT/pytest_gpu.log"; grep -n "^E  " "$OUT/pytest_gpu_full.log" | head -40
tail -4 "$OUT/pytest_gpu.log"
( timeout 900 python bench.py --steps 20 --warmup 5 2> "$OUT/bench.err" | tail -1 ) > "$OUT/bench_line.json"
cat "$OUT/bench_line.json"; tail -3 "$OUT/bench.err"
( timeout 600 python bench.py --steps 20 --warmup 5 --timing-phases 1 --no-cpu-baseline --as-written 0 2> "$OUT/bench_sync.err" | tail -1 ) > "$OUT/bench_line_sync.json"
cat "$OUT/bench_line_sync.json"
bash scripts/pmc_sq.sh $TAG/sq --steps 4 --warmup 2 --preroll 4 --check-channels 0 --as-written 0 > "$OUT/sq.log" 2>&1
tail -45 "$OUT/sq.log"
du -sh "$OUT"
