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
if has newtests; then timeout 900 python -m pytest tests/test_burst_recording.py -m gpu -q --tb=short 2>&1 | tail -15 | tee "$OUT/pytest_new.log"; fi
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
