#!/bin/bash
TAG=${1:-lazy2}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
B="--steps 8 --warmup 3 --no-cpu-baseline --as-written 0 --check-channels 16 --no-other-workloads --sustain 0 --no-state"
run() { JAERO_HIP_LIB=$2 timeout 300 python bench.py --workload msk $B $3 2> "$OUT/err_$1.txt" | tail -1 > "$OUT/line_$1.json"
  python - "$OUT/line_$1.json" "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], json.dumps(d['config'].get('kernel_ms_per_step') or {}), (d['config'].get('oracle_check') or {}).get('max_soft_byte_diff'), (d['config'].get('oracle_check') or {}).get('hard_bits_equal'))
PY
}
for rep in 1 2; do for l in gpurun_tmp/*.so; do n=$(basename $l .so); run "600_${n#libjaero_hip_}_$rep" $l "--fb 600"; done; done
for l in gpurun_tmp/*.so; do n=$(basename $l .so); run "1200_${n#libjaero_hip_}" $l ""; done
