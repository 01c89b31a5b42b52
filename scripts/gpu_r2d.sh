#!/bin/bash
# round 2, call D: whole GPU suite, kernel stats + SQ counters of the default bench
set -u
TAG=${1:-r2d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q --durations=8 --tb=short > "$OUT/pytest_gpu_full.log" 2>&1; tail -15 "$OUT/pytest_gpu_full.log" > "$OUT/pytest_gpu.log"; grep -n "^E  \|^FAILED\|passed\|failed" "$OUT/pytest_gpu_full.log" | head -40
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o stats -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --preroll 8 --no-cpu-baseline --as-written 0 --check-channels 0 > "$OUT/bench_prof_line.json" 2> "$OUT/prof.err"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && grep -E "k_oqpsk|k_coarse|Name" "$f" | cut -c1-200
cd "$GRAFT_REPO_ROOT"
bash scripts/pmc_sq.sh $TAG/sq --steps 6 --warmup 2 --preroll 40 --check-channels 0 --as-written 0 > "$OUT/sq.log" 2>&1
python - "$OUT/sq/sq_summary.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    w=v.get('SQ_WAVES',1); 
    print(k[:50], {c:round(x/ (w*4096) ,1) if c.startswith('SQ_INSTS') else round(x/v['SQ_WAVE_CYCLES'],3) if c in('SQ_WAIT_ANY','SQ_ACTIVE_INST_ANY','SQ_ACTIVE_INST_VALU','SQ_WAIT_INST_ANY','SQ_WAIT_INST_LDS','SQ_ACTIVE_INST_LDS') else x for c,x in v.items()})
PY
find "$OUT" -name "*.csv" -size +4M -delete
du -sh "$OUT"
