#!/bin/bash
# Round 5's GPU experiments, one stage each (the round's EVIDENCE pass is scripts/gpu_evidence.sh): the GPU suite, the driver's bench command, the
# whole recording from five starting points (product and A/B builds: make -C jaero_amd/csrc ab), the cost of the exact arithmetic (variants), the
# phase traces (make trace), the sample loop at bank sizes 1024 .. 32768 (sizes), burst kernels op for op against fused (burst_ab), and the
# small-bank A/Bs.  What each decided is in profiles/r5_*.md / .json and DESIGN 9 items 18-21.
# usage: scripts/gpu_r5.sh <tag> [what...]   what: tests bench recording recording_ab variants trace sizes
set -u
TAG=${1:-r5a}; shift || true
WHAT=${*:-tests bench recording sizes}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
cd "$R"
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=15 --tb=short > "$OUT/pytest_gpu_full.log" 2>&1
  tail -30 "$OUT/pytest_gpu_full.log" > "$OUT/pytest_gpu.log"; grep -n "^E  \|^FAILED\|passed\|failed" "$OUT/pytest_gpu_full.log" | head -30
  cp gpurun_out/soft_byte_ledger.json "$OUT/soft_byte_ledger.json" 2>/dev/null
fi
if has bench; then
  SECONDS=0; ( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> "$OUT/bench.err" | tail -1 ) > "$OUT/bench_line.json"
  echo "driver bench wall: ${SECONDS}s" | tee "$OUT/bench_wall.txt"; cut -c1-400 "$OUT/bench_line.json"; echo; tail -2 "$OUT/bench.err"
fi
if has recording; then
  ( timeout 900 python scripts/recording_full.py gpu 2> "$OUT/recording_full.err" | tail -1 ) > "$OUT/recording_full_gpu.json"; cut -c1-600 "$OUT/recording_full_gpu.json"; echo
fi
if has recording_ab; then
  ( JAERO_HIP_LIB=$R/gpurun_tmp/libjaero_hip_libm.so timeout 900 python scripts/recording_full.py gpu 2> "$OUT/recording_full_libm.err" | tail -1 ) > "$OUT/recording_full_gpu_device_libm.json"; cut -c1-600 "$OUT/recording_full_gpu_device_libm.json"; echo
  # the same A/B on the bench's checked channels and its kernel times
  ( JAERO_HIP_LIB=$R/gpurun_tmp/libjaero_hip_libm.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --as-written 0 2> "$OUT/bench_libm.err" | tail -1 ) > "$OUT/bench_line_device_libm.json"; cut -c1-300 "$OUT/bench_line_device_libm.json"; echo
fi
if has variants; then
  # sample-loop cost of the exact functions: the product against A/B builds that call the device library's atan2 (libatan2) or atan2 and hypot (libm)
  for v in product libatan2 libm; do
    L=$R/jaero_amd/libjaero_hip.so; [ $v != product ] && L=$R/gpurun_tmp/libjaero_hip_$v.so
    ( JAERO_HIP_LIB=$L timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --check-channels 0 2> "$OUT/bench_var_$v.err" | tail -1 ) > "$OUT/bench_line_var_$v.json"
    python - "$OUT/bench_line_var_$v.json" $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d["config"]
print(sys.argv[2], d["value"], d["ms_per_step"], c.get("kernel_ms_per_step"), {k:(v["msamples_per_s"], v["ms_per_step"]) for k,v in (c.get("as_written") or {}).items() if isinstance(v,dict)})
PY
  done
fi
if has trace; then
  for n in 65536 4096; do
    JAERO_HIP_LIB=$R/gpurun_tmp/libjaero_hip_trace.so timeout 600 python bench.py --channels $n --steps 12 --warmup 4 --no-cpu-baseline --as-written 0 --check-channels 0 > "$OUT/trace_bench_$n.json" 2> "$OUT/trace_$n.err"
    grep fb_trace "$OUT/trace_$n.err" | tail -1 > "$OUT/fb_trace_$n.json"; cat "$OUT/fb_trace_$n.json"
  done
fi
if has workloads; then
  for wl in msk burst_oqpsk burst_msk aerol aerol_burst aerol_c oqpsk8400; do
    extra=""; [ $wl = oqpsk8400 ] && extra="--as-written 0"
    ( timeout 600 python bench.py --workload $wl $extra --no-cpu-baseline 2> "$OUT/bench_$wl.err" | tail -1 ) > "$OUT/bench_line_$wl.json"
    python - "$OUT/bench_line_$wl.json" $wl <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print(sys.argv[2], d["value"], d["unit"], d["ms_per_step"], c.get("kernel_ms_per_step"), (c.get("oracle_check") or {}).get("max_soft_byte_diff"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    tail -2 "$OUT/bench_$wl.err"
  done
fi
if has burst_ab; then
  # burst kernels op for op (product) against an A/B build whose burst kernels keep the fused matched filter and the device library's hypot / atan2
  for v in product burstfast; do
    L=$R/jaero_amd/libjaero_hip.so; [ $v != product ] && L=$R/gpurun_tmp/libjaero_hip_$v.so
    ( JAERO_HIP_LIB=$L timeout 600 python scripts/burst_ab.py 2> "$OUT/burst_ab_$v.err" | tail -1 ) > "$OUT/burst_ab_$v.json"; cat "$OUT/burst_ab_$v.json"; echo
    for wl in burst_oqpsk burst_msk; do
      ( JAERO_HIP_LIB=$L timeout 600 python bench.py --workload $wl --no-cpu-baseline 2> "$OUT/bench_${wl}_$v.err" | tail -1 ) > "$OUT/bench_line_${wl}_$v.json"
      python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], sys.argv[3], d['value'], d['ms_per_step'])" "$OUT/bench_line_${wl}_$v.json" $wl $v
    done
  done
fi
if has msk600; then
  for v in product msk600tb40; do
    L=$R/jaero_amd/libjaero_hip.so; [ $v != product ] && L=$R/gpurun_tmp/libjaero_hip_$v.so
    echo "msk600 $v: $(JAERO_HIP_LIB=$L timeout 300 python scripts/time_msk600.py 65536 2>/dev/null | tail -1)" | tee -a "$OUT/msk600_ab.txt"
  done
fi
if has solo; then
  for v in product solod12; do
    L=$R/jaero_amd/libjaero_hip.so; [ $v != product ] && L=$R/gpurun_tmp/libjaero_hip_$v.so
    for n in 4096 16384; do
      ( JAERO_HIP_LIB=$L timeout 300 python bench.py --channels $n --steps 12 --warmup 4 --no-cpu-baseline --as-written 0 --check-channels 0 2>/dev/null | tail -1 ) > "$OUT/bench_line_${n}_$v.json"
      python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], sys.argv[3], d['value'], d['ms_per_step'], d['config']['kernel_ms_per_step'])" "$OUT/bench_line_${n}_$v.json" $n $v | tee -a "$OUT/solo_ab.txt"
    done
  done
fi
if has burstworst; then
  ( timeout 600 python scripts/burst_ab.py 2> "$OUT/burst_worst.err" | tail -1 ) > "$OUT/burst_worst.json"; cat "$OUT/burst_worst.json"; echo
fi
if has msktrace; then
  for n in 65536 256; do
    JAERO_HIP_LIB=$R/gpurun_tmp/libjaero_hip_trace.so timeout 600 python bench.py --workload msk --channels $n --steps 12 --warmup 4 --no-cpu-baseline --as-written 0 --check-channels 0 > "$OUT/msk_trace_bench_$n.json" 2> "$OUT/msk_trace_$n.err"
    grep fb_trace "$OUT/msk_trace_$n.err" | tail -1 > "$OUT/msk_fb_trace_$n.json"; cat "$OUT/msk_fb_trace_$n.json"
  done
fi
if has recording_atan2; then
  ( JAERO_HIP_LIB=$R/gpurun_tmp/libjaero_hip_libatan2.so timeout 900 python scripts/recording_full.py gpu 2> "$OUT/recording_full_libatan2.err" | tail -1 ) > "$OUT/recording_full_gpu_device_atan2.json"; cut -c1-400 "$OUT/recording_full_gpu_device_atan2.json"; echo
fi
if has big; then
  ( timeout 600 python bench.py --channels 131072 --steps 10 --warmup 3 --no-cpu-baseline --as-written 0 2> "$OUT/bench_131072.err" | tail -1 ) > "$OUT/bench_line_131072_channels.json"; cut -c1-200 "$OUT/bench_line_131072_channels.json"; echo
fi
if has msksmall; then
  for v in product ${MSK_SMALL_VARIANTS:-}; do   # MSK_SMALL_VARIANTS="msk1tb24 msk1tb36": A/B builds with -DMFB1_TB=.. in gpurun_tmp/
    L=$R/jaero_amd/libjaero_hip.so; [ $v != product ] && L=$R/gpurun_tmp/libjaero_hip_$v.so
    for n in 256 4096; do
      ( JAERO_HIP_LIB=$L timeout 300 python bench.py --workload msk --channels $n --steps 12 --warmup 4 --no-cpu-baseline --as-written 0 --check-channels 0 2>/dev/null | tail -1 ) > "$OUT/bench_line_msk_${n}_$v.json"
      python -c "import json,sys; d=json.load(open(sys.argv[1])); print('msk', sys.argv[2], sys.argv[3], d['value'], d['ms_per_step'], d['config']['kernel_ms_per_step'])" "$OUT/bench_line_msk_${n}_$v.json" $n $v | tee -a "$OUT/msk_small_ab.txt"
    done
  done
  [ -z "${SKIP_MSK_TESTS:-}" ] && timeout 900 python -m pytest tests -m gpu -q -k "msk" 2>&1 | tail -3
fi
if has sizes; then
  for n in 1024 4096 16384 32768; do
    ( timeout 300 python bench.py --channels $n --steps 12 --warmup 4 --no-cpu-baseline --as-written 0 --check-channels 0 2> "$OUT/bench_$n.err" | tail -1 ) > "$OUT/bench_line_${n}_channels.json"
    python - "$OUT/bench_line_${n}_channels.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d["config"]
print(c["channels_per_gpu"], d["value"], d["ms_per_step"], c.get("kernel_ms_per_step"))
PY
  done
fi
du -sh "$OUT"
