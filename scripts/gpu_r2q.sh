#!/bin/bash
# 8400 bps prefilter by overlap-save FFT (k_pre8400_fft): parity tests, then the workload with both forms of the filter
set -u
TAG=${1:-r2q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests -m gpu -q -k "8400 or fastfir or prefilter or aerol_c" --tb=short > "$OUT/pytest_8400.log" 2>&1; tail -15 "$OUT/pytest_8400.log"
B="--workload oqpsk8400 --steps 6 --warmup 2 --no-cpu-baseline --as-written 0"
( timeout 600 python bench.py $B 2> "$OUT/bench_fft.err" | tail -1 ) > "$OUT/bench_fft.json"
( JAERO_PRE8400=direct timeout 600 python bench.py $B --check-channels 0 2> "$OUT/bench_direct.err" | tail -1 ) > "$OUT/bench_direct.json"
python - "$OUT" <<'PY'
import json, sys
for n in ("bench_fft", "bench_direct"):
    try:
        d = json.loads(open(f"{sys.argv[1]}/{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["config"]["kernel_ms_per_step"], d["config"].get("oracle_check"))
    except Exception as e:
        print(n, "ERR", e)
PY
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o stats -- python "$GRAFT_REPO_ROOT/bench.py" $B --check-channels 0 > "$OUT/bench_prof_line.json" 2> "$OUT/prof.err"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "^\"Name|k_" "$f" | cut -c1-220 > "$OUT/kernel_stats.csv"; cat "$OUT/kernel_stats.csv"
find "$OUT" -name "*.csv" -size +6M -delete
