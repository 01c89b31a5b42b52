#!/bin/bash
# Round 6's GPU experiments, one stage each (the round's EVIDENCE pass stays scripts/gpu_evidence.sh).
# usage: scripts/gpu_r6.sh <tag> [what...]   what: probe tests bench wltime boxrow powercap trace ...
set -u
TAG=${1:-r6a}; shift || true
WHAT=${*:-probe tests bench}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
cd "$R"
if has probe; then bash scripts/diag/gpu_state_probe.sh > "$OUT/gpu_state_probe.txt" 2>&1; head -c 3000 "$OUT/gpu_state_probe.txt"; fi
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=15 --tb=short > "$OUT/pytest_gpu_full.log" 2>&1
  tail -30 "$OUT/pytest_gpu_full.log" > "$OUT/pytest_gpu.log"; grep -n "^E  \|^FAILED\|passed\|failed" "$OUT/pytest_gpu_full.log" | head -30
  cp gpurun_out/soft_byte_ledger.json "$OUT/soft_byte_ledger.json" 2>/dev/null
fi
if has bench; then
  SECONDS=0; ( timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 2> "$OUT/bench.err" | tail -1 ) > "$OUT/bench_line.json"
  echo "driver bench wall: ${SECONDS}s" | tee "$OUT/bench_wall.txt"; cut -c1-600 "$OUT/bench_line.json"; echo; tail -2 "$OUT/bench.err"
  cp gpurun_out/bench_details.json "$OUT/bench_details.json" 2>/dev/null
fi
if has wltime; then
  # how long does each of the other workloads take with six timed steps?  (sizing of bench.py's other_workloads pass)
  for wl in msk burst_oqpsk burst_msk aerol aerol_burst aerol_c oqpsk8400; do
    SECONDS=0; ( timeout 600 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --as-written 0 2> "$OUT/wl_$wl.err" | tail -1 ) > "$OUT/wl_line_$wl.json"
    echo "$wl wall ${SECONDS}s $(cut -c1-160 "$OUT/wl_line_$wl.json")" | tee -a "$OUT/wltime.txt"
  done
fi
if has burstab; then
  # the whole burst recording (staged by scripts/burst_recording_ab.py stage) through the product and through the A/B build of the burst OQPSK kernel
  ( timeout 900 python scripts/burst_recording_ab.py gpu 2> "$OUT/burst_rec_product.err" | tail -1 ) > "$OUT/burst_recording_gpu_product.json"; cut -c1-500 "$OUT/burst_recording_gpu_product.json"; echo
  ( JAERO_HIP_LIB=$R/gpurun_tmp/libjaero_hip_burstexact.so timeout 900 python scripts/burst_recording_ab.py gpu 2> "$OUT/burst_rec_exact.err" | tail -1 ) > "$OUT/burst_recording_gpu_exact.json"; cut -c1-500 "$OUT/burst_recording_gpu_exact.json"; echo
  python - "$OUT" <<'PY'
import json, sys
for n in ("product", "exact"):
    try:
        d = json.load(open(f"{sys.argv[1]}/burst_recording_gpu_{n}.json")); print(n, d["lib"], d["total"])
    except Exception as e: print(n, "failed", e)
PY
  # what the exact arithmetic costs on the burst workload
  for v in product exact; do
    L=$R/jaero_amd/libjaero_hip.so; [ $v = exact ] && L=$R/gpurun_tmp/libjaero_hip_burstexact.so
    ( JAERO_HIP_LIB=$L timeout 600 python bench.py --workload burst_oqpsk --steps 12 --warmup 2 --no-cpu-baseline --as-written 0 --check-channels 0 2> "$OUT/bench_burst_$v.err" | tail -1 ) > "$OUT/bench_line_burst_$v.json"
    python -c "import json;d=json.load(open('$OUT/bench_line_burst_$v.json'));print('$v',d['value'],d['ms_per_step'],d['config'].get('kernel_ms_total'))"
  done
fi
if has msk600; then
  timeout 1200 python -m pytest tests -m gpu -q -k "msk or Msk or MSK or golden or flags or family" --tb=short 2>&1 | tail -8 | tee "$OUT/pytest_msk.log"
  for args in "--fb 600" "--fb 1200"; do
    ( timeout 600 python bench.py --workload msk $args --steps 12 --warmup 4 --no-cpu-baseline --as-written 0 2> "$OUT/bench_msk.err" | tail -1 ) > "$OUT/bench_line_msk.json"
    python -c "import json;d=json.load(open('$OUT/bench_line_msk.json'));c=d['config'];print('$args',d['value'],d['ms_per_step'],c['kernel_ms_per_step'],c.get('oracle_check',{}).get('hard_bits_equal'),c.get('oracle_check',{}).get('max_soft_byte_diff'))"
    cp "$OUT/bench_line_msk.json" "$OUT/bench_line_msk_$(echo $args | tr -d ' -').json"
  done
fi
if has burstsplit; then
  timeout 1200 python -m pytest tests -m gpu -q -k "burst or Burst or qt or recording" --tb=short 2>&1 | tail -12 | tee "$OUT/pytest_burst.log"
  ( timeout 600 python bench.py --workload burst_oqpsk --steps 12 --warmup 2 --no-cpu-baseline --as-written 0 2> "$OUT/bench_burst.err" | tail -1 ) > "$OUT/bench_line_burst_oqpsk.json"
  python -c "import json;d=json.load(open('$OUT/bench_line_burst_oqpsk.json'));c=d['config'];print('burst_oqpsk',d['value'],d['ms_per_step'],c.get('kernel_ms_total'),c.get('oracle_check'))"
  ( timeout 600 python scripts/burst_recording_ab.py gpu 2> "$OUT/burst_rec.err" | tail -1 ) > "$OUT/burst_recording_gpu.json"; python -c "import json;d=json.load(open('$OUT/burst_recording_gpu.json'));print(d['total'])"
fi
if has pre8400; then
  timeout 1200 python -m pytest tests -m gpu -q -k "8400" --tb=short 2>&1 | tail -6 | tee "$OUT/pytest_8400.log"
  ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof8400" -o stats -- python bench.py --workload oqpsk8400 --preroll 40 --steps 6 --warmup 2 --no-cpu-baseline --as-written 0 --sustain 0 --no-state 2> "$OUT/bench_8400.err" | tail -1 ) > "$OUT/bench_line_8400.json"
  python -c "import json;d=json.load(open('$OUT/bench_line_8400.json'));c=d['config'];print('oqpsk8400',d['value'],d['ms_per_step'],c['kernel_ms_per_step'],c.get('oracle_check',{}).get('hard_bits_equal'),c.get('oracle_check',{}).get('max_soft_byte_diff'))"
  f=$(find "$OUT/prof8400" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_8400.csv" && grep -E "k_|Name" "$f" | cut -c1-160 | head -8
fi
if has aerolp; then
  timeout 1200 python -m pytest tests -m gpu -q -k "aerol or recording" --tb=short 2>&1 | tail -8 | tee "$OUT/pytest_aerol.log"
  for wl in aerol aerol_c aerol_burst; do
    ( timeout 600 python bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --as-written 0 2> "$OUT/bench_$wl.err" | tail -1 ) > "$OUT/bench_line_$wl.json"
    python -c "import json;d=json.load(open('$OUT/bench_line_$wl.json'));c=d['config'];print('$wl',d['value'],d['ms_per_step'],c.get('kernel_ms_per_step'),str(c.get('oracle_check'))[:160])"
  done
fi
if has tb600; then
  for v in product tb52 tb44; do
    L=$R/jaero_amd/libjaero_hip.so; [ $v != product ] && L=$R/gpurun_tmp/libjaero_hip_$v.so
    ( JAERO_HIP_LIB=$L timeout 600 python bench.py --workload msk --fb 600 --steps 12 --warmup 4 --no-cpu-baseline --as-written 0 --no-state 2> "$OUT/bench_tb.err" | tail -1 ) > "$OUT/bench_line_msk600_$v.json"
    python -c "import json;d=json.load(open('$OUT/bench_line_msk600_$v.json'));c=d['config'];print('$v',d['value'],d['ms_per_step'],c['kernel_ms_per_step'],c.get('oracle_check',{}).get('hard_bits_equal'))" | tee -a "$OUT/tb600.txt"
  done
fi
if has trace600; then
  # phase trace of the 600 bps MSK pair kernel (make -C jaero_amd/csrc trace): where does a sample's 4 us go?
  for args in "--fb 600 --channels 65536" "--fb 600 --channels 32768" "--fb 1200 --channels 65536"; do
    ( JAERO_HIP_LIB=$R/gpurun_tmp/libjaero_hip_trace.so timeout 600 python bench.py --workload msk $args --steps 12 --warmup 4 --no-cpu-baseline --as-written 0 --check-channels 0 --no-state 2> "$OUT/trace_msk.err" | tail -1 | cut -c1-200 ) ; grep fb_trace "$OUT/trace_msk.err" | tail -1 | tee -a "$OUT/trace_msk.jsonl"
  done
fi
if has divcheck; then ( ./scripts/ubench/div_check 4096 | tee "$OUT/div_check.json" ); fi
if has boxrow; then
  # one row of profiles/r6_box_table.md: the driver's command on this (fresh) box, headline against calibration and device state
  ( timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 2> "$OUT/boxrow.err" | tail -1 ) > "$OUT/boxrow_line.json"
  python - "$OUT/boxrow_line.json" <<'PY' | tee "$OUT/boxrow.json"
import json, sys
d = json.load(open(sys.argv[1])); c = d["config"]; g = d.get("gpu_state", {}); k = d.get("calib", {})
ow = c.get("other_workloads", {})
print(json.dumps({"value": d["value"], "ms_per_step": d["ms_per_step"], "step_ms": d.get("step_ms"), "kernel_ms_per_step": c["kernel_ms_per_step"], "frac": d["roofline"]["frac"],
                  "frac_of_calib_hbm": d["roofline"].get("frac_of_calib_hbm"), "calib": {x: k.get(x) for x in ("fp64_tflops", "hbm_gbs")}, "calib_cold": k.get("cold"),
                  "gpu_state": {x: g.get(x) for x in ("sclk_mhz_mean", "sclk_mhz_min", "power_w_mean", "power_w_max", "power_cap_w", "temp_hotspot_c_max", "throttle_residency", "throttled", "value_per_watt")},
                  "sustain": g.get("sustain"), "other": {n: [v.get("value"), v.get("sclk_mhz"), v.get("power_w")] for n, v in ow.items() if isinstance(v, dict)},
                  "cpu_baseline": (d.get("cpu_baseline") or {}).get("value")}))
PY
fi
if has newtests; then timeout 900 python -m pytest tests/test_burst_recording.py tests/test_gpu_parity.py -k "recording or silent_lanes" -m gpu -q --tb=short 2>&1 | tail -15 | tee "$OUT/pytest_new.log"; fi
if has sqaerol; then
  # VALU instruction counts of the Aero-L workloads (their roofline block is priced against VALU issue): STEPS_TOTAL = warm-up + timed (+ 4 re-written steps in aerol_burst)
  for wl in aerol aerol_burst aerol_c; do
    ST=8; [ $wl = aerol_burst ] && ST=12
    STEPS_TOTAL=$ST SQ_TAG="${TAG}_$wl" bash scripts/pmc_sq.sh "$TAG/sq_$wl" --workload $wl --steps 6 --warmup 2 --as-written 0 --check-channels 0 --no-state > "$OUT/sq_$wl.log" 2>&1
    cp "$OUT/sq_$wl/sq_summary.json" "$OUT/sq_summary_$wl.json" 2>/dev/null && cp "$OUT/sq_summary_$wl.json" "$R/profiles/sq_summary_$wl.json"
    python -c "import json;d=json.load(open('$OUT/sq_summary_$wl.json'));print('$wl',{k:round(v.get('SQ_INSTS_VALU_sum',0)/d.get('steps_total',1)) for k,v in d.items() if isinstance(v,dict)})"
  done
fi
du -sh "$OUT"
