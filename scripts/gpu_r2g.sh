#!/bin/bash
# diagnostic: timing of k_oqpsk_fb with parts switched off (JAERO_FB_DBG bits: 1 F no carrier gather, 2 B no symbol-NCO gather, 4 B no atan2 / no symbol instants, 8 F no matched filter)
set -u
TAG=${1:-r2g}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for dbg in 0 1 2 3 4 8 12 15; do
  ( JAERO_FB_DBG=$dbg timeout 300 python bench.py --steps 8 --warmup 2 --preroll 2 --no-cpu-baseline --as-written 0 --check-channels 0 2>/dev/null | tail -1 ) > "$OUT/bench_dbg$dbg.json"
  python -c "import json;d=json.load(open('$OUT/bench_dbg$dbg.json'));print('dbg $dbg',d['config']['kernel_ms_per_step'])"
done
