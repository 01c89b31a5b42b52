#!/bin/bash
# A/B of the lane skew in k_oqpsk_fb: same box, same library, JAERO_FB_SKEW=0 against the default; with --timing-phases 1 (all
# channels symbol-synchronous) as the bound of what aligning the lanes can give.   usage: scripts/diag/skew_ab.sh <tag> [tests]
TAG=${1:-skew}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
B="--steps 12 --warmup 4 --no-cpu-baseline --as-written 0 --check-channels 16 --no-other-workloads --sustain 1 --no-state"
if [[ " $* " == *" tests "* ]]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_recording.py tests/test_c_abi.py -m gpu -x -q 2>&1 | tail -5 | tee "$OUT/pytest.log"
fi
for rep in 1 2; do
for v in 1 0; do
  JAERO_FB_SKEW=$v timeout 300 python bench.py $B 2> "$OUT/err_$v.txt" | tail -1 > "$OUT/line_skew${v}_$rep.json"
  python - "$OUT/line_skew${v}_$rep.json" "skew=$v rep=$rep" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d.get('kernels') or d['config'].get('kernels') or {}
print(sys.argv[2], d['value'], d['ms_per_step'], json.dumps(d.get('kernel_ms') or d['config'].get('kernel_ms_per_step') or {}), d['config'].get('oracle_check'))
PY
done; done
JAERO_FB_SKEW=0 timeout 300 python bench.py $B --timing-phases 1 2>/dev/null | tail -1 > "$OUT/line_sync_skew0.json"
JAERO_FB_SKEW=1 timeout 300 python bench.py $B --timing-phases 1 2>/dev/null | tail -1 > "$OUT/line_sync_skew1.json"
for f in sync_skew0 sync_skew1; do python - "$OUT/line_$f.json" $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], json.dumps(d.get('kernel_ms') or d['config'].get('kernel_ms_per_step') or {}))
PY
done
tail -3 "$OUT"/err_*.txt
