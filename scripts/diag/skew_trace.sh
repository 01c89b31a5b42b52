#!/bin/bash
# phase traces (FB_TRACE, profiles/r5_fb_trace.md) of the sample loop with and without the lane skew, and of the library before it
TAG=${1:-skewtr}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
B="--steps 12 --warmup 4 --no-cpu-baseline --as-written 0 --check-channels 0 --no-other-workloads --sustain 0 --no-state"
run() { # name lib skew extra
  JAERO_HIP_LIB=$2 JAERO_FB_SKEW=$3 timeout 300 python bench.py $B $4 2> "$OUT/err_$1.txt" | tail -1 > "$OUT/line_$1.json"
  python - "$OUT/line_$1.json" "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], json.dumps(d['config'].get('kernel_ms_per_step') or {}))
PY
  grep -h "fb_trace" "$OUT/err_$1.txt" | tail -1 | cut -c1-900
}
run old_unsync gpurun_tmp/libjaero_hip_old.so 0 ""
run old_sync gpurun_tmp/libjaero_hip_old.so 0 "--timing-phases 1"
run oldtr_unsync gpurun_tmp/libjaero_hip_old_trace.so 0 ""
run oldtr_sync gpurun_tmp/libjaero_hip_old_trace.so 0 "--timing-phases 1"
run newtr_skew0 gpurun_tmp/libjaero_hip_trace.so 0 ""
run newtr_skew1 gpurun_tmp/libjaero_hip_trace.so 1 ""
run newtr_sync0 gpurun_tmp/libjaero_hip_trace.so 0 "--timing-phases 1"
