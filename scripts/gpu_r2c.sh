#!/bin/bash
# round 2, call C: front/back pair kernel -- parity under both pair counts, then the default bench with it and with the single-wavefront kernel
set -u
TAG=${1:-r2c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
T="tests/test_gpu_parity.py tests/test_gpu_scale.py::test_oqpsk_4096_channels tests/test_gpu_scale.py::test_10500_and_8400_banks_alive_together tests/test_gpu_ingest.py tests/test_gpu_aerol.py::test_pcm_to_signal_units_on_device"
for mode in pairs4 pairs1; do
  JAERO_OQPSK_KERNEL=$mode timeout 900 python -m pytest $T -m gpu -q --tb=short -x > "$OUT/pytest_$mode.log" 2>&1
  tail -3 "$OUT/pytest_$mode.log"; grep -n "^E  " "$OUT/pytest_$mode.log" | head -20
done
for mode in auto; do
  ( JAERO_OQPSK_KERNEL=$mode timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --as-written 0 2> "$OUT/bench_$mode.err" | tail -1 ) > "$OUT/bench_$mode.json"
  python - "$OUT/bench_$mode.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], c["kernel_ms_per_step"], c.get("oracle_check",{}).get("hard_bits_equal"), c.get("oracle_check",{}).get("max_soft_byte_diff"), c.get("oracle_check",{}).get("ber_gpu"))
except Exception as e:
    print("bench parse failed", e)
PY
  tail -3 "$OUT/bench_$mode.err"
done
( JAERO_BENCH_CHANNELS=4096 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --as-written 0 --check-channels 4 --preroll 0 2>/dev/null | tail -1 ) > "$OUT/bench_4096.json"
python -c "import json;d=json.load(open('$OUT/bench_4096.json'));print('4096ch',d['value'],d['config']['kernel_ms_per_step'])"
du -sh "$OUT"
