#!/bin/bash
# A/B of k_oqpsk_fb orderings (variant libraries through JAERO_HIP_LIB)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r2i}; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in "" _vB _vC; do
  ( JAERO_HIP_LIB=$GRAFT_REPO_ROOT/jaero_amd/libjaero_hip$v.so timeout 200 python bench.py --steps 10 --warmup 3 --preroll 30 --no-cpu-baseline --as-written 0 --check-channels 2 2>/dev/null | tail -1 ) > "$OUT/bench$v.json"
  python -c "import json;d=json.load(open('$OUT/bench$v.json'));print('variant [$v]',d['config']['kernel_ms_per_step'],d['config'].get('oracle_check',{}).get('hard_bits_equal'))"
done
