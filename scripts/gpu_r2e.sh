#!/bin/bash
# round 2, call E: pair-count experiment at 65536 channels, burst-bank test re-run
set -u
TAG=${1:-r2e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -q --tb=short > "$OUT/pytest_scale.log" 2>&1; tail -3 "$OUT/pytest_scale.log"; grep -n "^E  " "$OUT/pytest_scale.log" | head
for mode in pairs1 pairs4; do
  ( JAERO_OQPSK_KERNEL=$mode timeout 600 python bench.py --steps 12 --warmup 3 --preroll 30 --no-cpu-baseline --as-written 0 --check-channels 4 2> "$OUT/bench_$mode.err" | tail -1 ) > "$OUT/bench_$mode.json"
  python -c "import json;d=json.load(open('$OUT/bench_$mode.json'));c=d['config'];print('$mode',d['value'],c['kernel_ms_per_step'],c.get('oracle_check',{}).get('hard_bits_equal'))"
done
