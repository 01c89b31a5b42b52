#!/bin/bash
# parity subset for the coarse kernel + sample kernel, then the bench
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r2j}; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q --tb=short > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log"; grep -n "^E  " "$OUT/pytest.log" | head
( timeout 300 python bench.py --steps 10 --warmup 3 --preroll 30 --no-cpu-baseline --as-written 0 --check-channels 4 2>"$OUT/bench.err" | tail -1 ) > "$OUT/bench.json"
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['config']['kernel_ms_per_step'],d['config'].get('oracle_check',{}).get('hard_bits_equal'))"; tail -2 "$OUT/bench.err"
