#!/bin/bash
# One GPU-box pass that produces everything a round's evidence needs (replaces the per-experiment scripts of rounds 1-2):
#   smoke(), the whole GPU suite, the driver's bench command, rocprofv3 kernel stats, HBM counters (two --pmc passes) and SQ issue / wait
#   counters of a workload, the bench lines of the other workloads, a 2-rank line on this 1-GPU box (ranks share the device, flagged).
# usage: scripts/gpu_evidence.sh <tag> [what...]      what: smoke tests bench prof pmc sq workloads gpus2 (default: all)
#        PMC_WORKLOADS="oqpsk msk ..." selects the workloads the counter passes run on (default: oqpsk), WORKLOADS="aerol_c ..." those of the
#        `workloads` stage (default: all seven)
# Everything lands in gpurun_out/<tag>/; copy what is to be judged into profiles/ (scripts/collect_evidence.py <tag>).
set -u
TAG=${1:-evidence}; shift || true
WHAT=${*:-smoke tests prof pmc sq bench gpus2 gpus8}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
has() { [[ " $WHAT " == *" $1 "* ]]; }
cd "$R"
if has smoke; then ( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > "$OUT/smoke.log"; cat "$OUT/smoke.log"; fi
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=10 --tb=short > "$OUT/pytest_gpu_full.log" 2>&1
  tail -22 "$OUT/pytest_gpu_full.log" > "$OUT/pytest_gpu.log"; grep -n "^E  \|^FAILED\|passed\|failed" "$OUT/pytest_gpu_full.log" | head -20
fi
B="--steps 6 --warmup 2 --no-cpu-baseline --as-written 0 --check-channels 0 --no-other-workloads --sustain 0 --no-state"
for wl in ${PMC_WORKLOADS:-oqpsk}; do
  SFX=""; [ "$wl" != oqpsk ] && SFX="_$wl"
  PRE=""; case $wl in oqpsk|oqpsk8400) PRE="--preroll 40";; esac
  # the burst workloads' kernel times depend on where in its burst cycle a channel is: profile the steps the bench line times
  case $wl in burst_*|aerol*) B="--no-cpu-baseline --as-written 0 --check-channels 0 --no-state";; *) B="--steps 6 --warmup 2 --no-cpu-baseline --as-written 0 --check-channels 0 --no-other-workloads --sustain 0 --no-state";; esac
  KRE='k_oqpsk|k_msk|k_coarse|k_burst|k_hilbert|k_trident|k_pre8400|k_viterbi|k_aerol'
  cd /tmp
  if has prof; then
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof$SFX" -o stats -- python "$R/bench.py" --workload $wl $B $PRE > "$OUT/bench_prof_line$SFX.json" 2> "$OUT/prof$SFX.err"
    f=$(find "$OUT/prof$SFX" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats$SFX.csv" && grep -E "k_|Name" "$f" | cut -c1-200 | head -12
  fi
  if has pmc; then
    mkdir -p "$OUT/pmc$SFX"
    for c in WRITE_SIZE FETCH_SIZE; do
      timeout 400 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc $c --output-format csv -d "$OUT/pmc${SFX}_$c" -o pmc -- python "$R/bench.py" --workload $wl $B $PRE > "$OUT/pmc${SFX}_$c.log" 2>&1
      f=$(find "$OUT/pmc${SFX}_$c" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/pmc$SFX/pmc_$c.csv"
    done
    cd "$R"; python scripts/summarize_pmc.py "$OUT/pmc$SFX" "${TAG}_$wl" ${CHANNELS:-65536} "$OUT/pmc_summary$SFX.json" | head -5
  fi
  # steps of work the profiled command issues (the Aero-L lines price a whole step against VALU issue): warm-up + timed (+ 4 re-written steps in aerol_burst)
  ST=8; case $wl in aerol|aerol_c|burst_*) ST=36;; aerol_burst) ST=40;; esac
  if has sq; then cd "$R"; STEPS_TOTAL=$ST SQ_TAG="${TAG}_$wl" bash scripts/pmc_sq.sh "$TAG/sq$SFX" --workload $wl $B $PRE > "$OUT/sq$SFX.log" 2>&1; cp "$OUT/sq$SFX/sq_summary.json" "$OUT/sq_summary$SFX.json" 2>/dev/null; fi
done
cd "$R"
# the counter summaries of THIS pass become the ones bench.py reads, so that the lines below are annotated with traffic taken in the same pass
export JAERO_EVIDENCE_TAG=$TAG
for f in "$OUT"/pmc_summary*.json "$OUT"/sq_summary*.json; do [ -s "$f" ] && cp "$f" "$R/profiles/$(basename "$f")"; done
[ -s "$OUT/pmc_summary_oqpsk8400.json" ] && cp "$OUT/pmc_summary_oqpsk8400.json" "$R/profiles/pmc_summary_8400.json" # (the name bench.py reads for the 8400 bps workload)
if has bench; then
  rm -f gpurun_out/bench_details.json
  SECONDS=0; ( timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 2> "$OUT/bench.err" | tail -1 ) > "$OUT/bench_line.json"
  cp gpurun_out/bench_details.json "$OUT/bench_details.json" 2>/dev/null
  echo "driver bench wall: ${SECONDS}s" | tee "$OUT/bench_wall.txt"; cut -c1-300 "$OUT/bench_line.json"; echo; tail -2 "$OUT/bench.err"
fi
if has workloads; then
  for wl in ${WORKLOADS:-msk burst_oqpsk burst_msk aerol aerol_burst aerol_c oqpsk8400}; do
    extra=""; [ $wl = oqpsk8400 ] && extra="--as-written 0"
    ( timeout 600 python bench.py --workload $wl $extra 2> "$OUT/bench_$wl.err" | tail -1 ) > "$OUT/bench_line_$wl.json"; cut -c1-200 "$OUT/bench_line_$wl.json"; echo
  done
fi
if has gpus2; then ( timeout 600 python bench.py --gpus 2 --channels 16384 --steps 8 --warmup 3 --no-cpu-baseline --sustain 0.5 2> "$OUT/bench_gpus2.err" | tail -1 ) > "$OUT/bench_line_gpus2_shared_device.json"; cut -c1-200 "$OUT/bench_line_gpus2_shared_device.json"; echo; fi
if has gpus8; then
  # the driver's 8-GPU command on this 1-GPU lease: eight ranks share the device (gloo control plane, RCCL edge operations skipped, flagged in
  # the line) -- BASELINE configs[4] as written, 4096 channels per GPU
  ( timeout 900 python bench.py --gpus 8 --channels 4096 --steps 8 --warmup 3 --no-cpu-baseline --sustain 0.5 2> "$OUT/bench_gpus8.err" | tail -1 ) > "$OUT/bench_line_gpus8_shared_device.json"; cut -c1-300 "$OUT/bench_line_gpus8_shared_device.json"; echo; tail -3 "$OUT/bench_gpus8.err"
fi
find "$OUT" -name "*.csv" -size +6M -delete
find "$OUT" -type d -name "pmc*_SIZE" -prune -exec rm -rf {} + 2>/dev/null
du -sh "$OUT"
