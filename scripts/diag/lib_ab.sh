#!/bin/bash
# generic A/B of library builds: scripts/diag/lib_ab.sh <tag> <reps> "<workload args>" lib1.so lib2.so ...   (libraries relative to the repo root)
TAG=$1; REPS=$2; WL=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
B="--steps 10 --warmup 3 --no-cpu-baseline --as-written 0 --check-channels 16 --no-other-workloads --sustain 0 --no-state"
for rep in $(seq 1 $REPS); do for l in "$@"; do n=$(basename $l .so); n=${n#libjaero_hip}; n=${n#_}; [ -z "$n" ] && n=product
  JAERO_HIP_LIB=$l timeout 300 python bench.py $B $WL 2> "$OUT/err_$n.txt" | tail -1 > "$OUT/line_${n}_$rep.json"
  python - "$OUT/line_${n}_$rep.json" "$n/$rep" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); oc=d['config'].get('oracle_check') or {}
print(sys.argv[2], d['value'], d['ms_per_step'], json.dumps(d['config'].get('kernel_ms_per_step') or {}), oc.get('max_soft_byte_diff'), oc.get('hard_bits_equal'))
PY
done; done
