#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "8400 or pre8400 or prefilt" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pre}; mkdir -p $OUT
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o stats -- python "$GRAFT_REPO_ROOT/bench.py" --workload oqpsk8400 --steps 6 --warmup 2 --no-cpu-baseline --as-written 0 --check-channels 16 --no-other-workloads --sustain 0 --no-state --preroll 40 > "$OUT/line.json" 2> "$OUT/err.txt"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); grep -E "k_pre8400|k_coarse|k_oqpsk" "$f" | cut -c1-60,100-200
python -c "import json; d=json.load(open('$OUT/line.json')); print(d['value'], d['ms_per_step'], d['config'].get('oracle_check'))"
rm -rf "$OUT/prof"
