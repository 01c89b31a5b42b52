#!/bin/bash
# A/B of the back half's lazily shifted register tail in k_msk_fb (MFB_LAZY_K = 1 / 2 / 4 / 8; the product library is 4)
TAG=${1:-lazy}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
B="--steps 8 --warmup 3 --no-cpu-baseline --as-written 0 --check-channels 16 --no-other-workloads --sustain 0 --no-state"
if [[ " $* " == *" tests "* ]]; then
  timeout 1200 python -m pytest tests -m gpu -x -q -k "msk or Msk or MSK" 2>&1 | tail -4 | tee "$OUT/pytest.log"
fi
run() { JAERO_HIP_LIB=$2 timeout 300 python bench.py --workload msk $B $3 2> "$OUT/err_$1.txt" | tail -1 > "$OUT/line_$1.json"
  python - "$OUT/line_$1.json" "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], json.dumps(d['config'].get('kernel_ms_per_step') or {}), (d['config'].get('oracle_check') or {}).get('max_soft_byte_diff'), (d['config'].get('oracle_check') or {}).get('hard_bits_equal'))
PY
}
for rep in 1 2; do
for k in 1 2 4 8; do
  lib=gpurun_tmp/libjaero_hip_lazy$k.so; [ $k = 4 ] && lib=jaero_amd/libjaero_hip.so
  run "600_k${k}_$rep" $lib "--fb 600"
done; done
for k in 1 4; do lib=gpurun_tmp/libjaero_hip_lazy$k.so; [ $k = 4 ] && lib=jaero_amd/libjaero_hip.so; run "1200_k$k" $lib ""; done
for k in 1 4; do lib=gpurun_tmp/libjaero_hip_lazy$k.so; [ $k = 4 ] && lib=jaero_amd/libjaero_hip.so; run "1200_256ch_k$k" $lib "--channels 256"; done
