#!/bin/bash
# round-4 GPU passes: usage scripts/gpu_r4.sh <tag> <what...>   (what: ubench newtests aerolbench tests bench ...)
set -u
TAG=${1:-r4}; shift || true
WHAT=${*:-ubench newtests aerolbench}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
cd "$R"
if has ubench; then
  timeout 300 scripts/ubench/atan2_rates 64 > "$OUT/atan2_rates.txt" 2>&1; cat "$OUT/atan2_rates.txt"
fi
if has coarse; then
  timeout 300 scripts/ubench/coarse_notrace 65536 4 > "$OUT/coarse_ab.txt" 2>&1; cat "$OUT/coarse_ab.txt"
  timeout 300 scripts/ubench/coarse_trace 65536 2 > "$OUT/coarse_trace.txt" 2>&1; grep -v "per launch" "$OUT/coarse_trace.txt"
fi
if has newtests; then
  timeout 1200 python -m pytest tests/test_gpu_scale_aerol.py tests/test_gpu_scale.py -m gpu -q -k "65536_channels and aerol or ragged" --durations=8 --tb=short > "$OUT/pytest_new.log" 2>&1
  tail -25 "$OUT/pytest_new.log"
fi
if has aerolbench; then
  for wl in aerol aerol_burst aerol_c; do
    ( timeout 600 python bench.py --workload $wl --no-cpu-baseline 2> "$OUT/bench_$wl.err" | tail -1 ) > "$OUT/bench_line_$wl.json"; cut -c1-1500 "$OUT/bench_line_$wl.json"; echo; tail -3 "$OUT/bench_$wl.err"
  done
fi
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=10 --tb=short > "$OUT/pytest_gpu_full.log" 2>&1
  tail -22 "$OUT/pytest_gpu_full.log"
fi
if has bench; then
  ( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> "$OUT/bench.err" | tail -1 ) > "$OUT/bench_line.json"; cut -c1-600 "$OUT/bench_line.json"; echo; tail -2 "$OUT/bench.err"
fi
du -sh "$OUT"
if has abbench; then
  # A/B: this tree's library against gpurun_tmp/libjaero_hip_old.so (the build before the change under test), same box, same run
  for wl in oqpsk msk; do
    ( timeout 600 python bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --as-written 0 2> "$OUT/ab_new_$wl.err" | tail -1 ) > "$OUT/ab_new_$wl.json"
  done
  cp jaero_amd/libjaero_hip.so /tmp/libjaero_hip_new.so; cp gpurun_tmp/libjaero_hip_old.so jaero_amd/libjaero_hip.so
  for wl in oqpsk msk; do
    ( timeout 600 python bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --as-written 0 2> "$OUT/ab_old_$wl.err" | tail -1 ) > "$OUT/ab_old_$wl.json"
  done
  cp /tmp/libjaero_hip_new.so jaero_amd/libjaero_hip.so
  python - <<PY
import json
for wl in ("oqpsk","msk"):
    for v in ("new","old"):
        try:
            l=json.load(open("$OUT/ab_%s_%s.json"%(v,wl)))
            print(wl, v, l["value"], l["ms_per_step"], l["config"].get("kernel_ms_per_step") or l["config"].get("kernel_ms_total"), (l["config"].get("oracle_check") or {}))
        except Exception as e: print(wl, v, "ERR", e, open("$OUT/ab_%s_%s.err"%(v,wl)).read()[-400:])
PY
fi
if has burstmsk; then
  timeout 1200 python -m pytest tests -m gpu -q -k "burst" --tb=short > "$OUT/pytest_burst.log" 2>&1; tail -8 "$OUT/pytest_burst.log"
  ( timeout 600 python bench.py --workload burst_msk --no-cpu-baseline 2> "$OUT/bench_burst_msk.err" | tail -1 ) > "$OUT/bench_line_burst_msk.json"; cut -c1-400 "$OUT/bench_line_burst_msk.json"; echo; tail -2 "$OUT/bench_burst_msk.err"
  python - <<PY
import json
l=json.load(open("$OUT/bench_line_burst_msk.json")); print(l["value"], l["ms_per_step"], l["config"].get("kernel_ms_per_step"), l["config"].get("oracle_check"))
PY
fi
if has bench8400; then
  ( timeout 600 python bench.py --workload oqpsk8400 --as-written 0 --no-cpu-baseline 2> "$OUT/bench_8400.err" | tail -1 ) > "$OUT/bench_line_oqpsk8400.json"
  python - <<PY
import json
l=json.load(open("$OUT/bench_line_oqpsk8400.json")); print("8400:", l["value"], l["ms_per_step"], l["config"].get("kernel_ms_per_step"), l["config"].get("oracle_check"))
PY
fi
if has msk; then
  timeout 1200 python -m pytest tests -m gpu -q -k "msk and not burst" --tb=short > "$OUT/pytest_msk.log" 2>&1; tail -6 "$OUT/pytest_msk.log"
  ( timeout 600 python bench.py --workload msk --no-cpu-baseline 2> "$OUT/bench_msk.err" | tail -1 ) > "$OUT/bench_line_msk.json"; tail -2 "$OUT/bench_msk.err"
  python - <<PY
import json
l=json.load(open("$OUT/bench_line_msk.json")); print("msk:", l["value"], l["ms_per_step"], l["config"].get("kernel_ms_per_step"), l["config"].get("kernel_hbm_frac"), l["config"].get("oracle_check"))
PY
fi
if has calib; then
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/calib_$c" -o pmc -- "$R/scripts/ubench/hbm_counters" > "$OUT/calib_$c.log" 2>&1
    f=$(find "$OUT/calib_$c" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/calib_$c.csv"
  done
  cd "$R"
  python - <<PY
import csv, json, collections
moved = {"rd8_rows": 8589934592, "wr8_rows": 8589934592, "wr16_own": 17179869184, "rd16_stream": 17179869184, "wr16_stream": 17179869184, "rd8_rows_nt": 8589934592, "wr8_rows_nt": 8589934592, "rd16_stream_nt": 17179869184}
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    try:
        for r in csv.DictReader(open("$OUT/calib_%s.csv" % c)):
            k = r["Kernel_Name"].split("(")[0]
            if k in moved: res[k][c + "_KiB"] = float(r["Counter_Value"])
    except Exception as e: print("calib", c, e)
for k, v in res.items():
    v["bytes_moved"] = moved[k]
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        if c + "_KiB" in v: v[c + "_bytes_over_moved"] = round(v[c + "_KiB"] * 1024 / moved[k], 4)
json.dump({"what": "scripts/ubench/hbm_counters.hip under rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes), MI355X; round 4 adds the non-temporal rows", "kernels": res}, open("$OUT/counter_calibration.json", "w"), indent=1)
print(json.dumps(res, indent=0))
PY
fi
if has burstoqpsk; then
  timeout 1200 python -m pytest tests -m gpu -q -k "burst" --tb=short > "$OUT/pytest_burst.log" 2>&1; tail -4 "$OUT/pytest_burst.log"
  ( timeout 600 python bench.py --workload burst_oqpsk --no-cpu-baseline 2> "$OUT/bench_burst_oqpsk.err" | tail -1 ) > "$OUT/bench_line_burst_oqpsk.json"; tail -2 "$OUT/bench_burst_oqpsk.err"
  python - <<PY
import json
l=json.load(open("$OUT/bench_line_burst_oqpsk.json")); print("burst_oqpsk:", l["value"], l["ms_per_step"], {k:round(v/l["steps"],3) for k,v in l["config"].get("kernel_ms_total",{}).items()}, l["config"].get("oracle_check"))
PY
fi
if has t8400; then
  timeout 1200 python -m pytest tests -m gpu -q -k "8400 or msk_600 or jfastfir or rate_change" --tb=short > "$OUT/pytest_8400.log" 2>&1; tail -4 "$OUT/pytest_8400.log"
fi
if has prof8400; then
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_8400" -o stats -- python "$R/bench.py" --workload oqpsk8400 --steps 6 --warmup 2 --no-cpu-baseline --as-written 0 --check-channels 0 --preroll 40 > "$OUT/prof8400.json" 2> "$OUT/prof8400.err"
  f=$(find "$OUT/prof_8400" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "k_|Name" "$f" | cut -c1-170 | head -12
  cd "$R"
fi
