#!/bin/bash
# round 2, call M: whole GPU suite after the clean-up and the full-sector coarse ring fill; short bench; HBM counter passes
set -u
TAG=${1:-r2m}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q --durations=5 --tb=short > "$OUT/pytest_gpu_full.log" 2>&1; tail -12 "$OUT/pytest_gpu_full.log" > "$OUT/pytest_gpu.log"; grep -n "^E  \|^FAILED\|passed\|failed" "$OUT/pytest_gpu_full.log" | head -20
JAERO_OQPSK_KERNEL=pairs4 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py::test_oqpsk_4096_channels -m gpu -q --tb=short > "$OUT/pytest_pairs4.log" 2>&1; tail -2 "$OUT/pytest_pairs4.log"
B="--steps 6 --warmup 2 --preroll 40 --no-cpu-baseline --as-written 0 --check-channels 4"
( timeout 300 python bench.py $B 2>/dev/null | tail -1 ) > "$OUT/bench_short.json"
python -c "import json;d=json.load(open('$OUT/bench_short.json'));print(d['value'],d['config']['kernel_ms_per_step'],d['config'].get('oracle_check',{}).get('hard_bits_equal'))"
cd /tmp
KRE='k_oqpsk|k_coarse'
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc $c --output-format csv -d "$OUT/pmc_$c" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" $B > "$OUT/pmc_$c.log" 2>&1
  f=$(find "$OUT/pmc_$c" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/pmc_$c.csv"
done
cd "$GRAFT_REPO_ROOT"
python - "$OUT" <<'PY'
import csv,collections,sys
out=sys.argv[1]
for c in ('FETCH_SIZE','WRITE_SIZE'):
    r=collections.defaultdict(list)
    for row in csv.DictReader(open(f'{out}/pmc_{c}.csv')):
        r[row['Kernel_Name'].split('(')[0][:40]].append(float(row['Counter_Value']))
    for k,v in r.items():
        m=max(v); w=[x for x in v if x>=0.1*m]
        print(c,k,len(w),round(sum(w)/len(w)*1024/1e9,2),'GB (raw counter)')
PY
find "$OUT" -name "*.csv" -size +6M -delete
