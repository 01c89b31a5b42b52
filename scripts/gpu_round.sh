#!/bin/bash
# One GPU-box pass: parity tests, default bench, rocprofv3 kernel stats, then two PMC passes (FETCH_SIZE / WRITE_SIZE
# separately, kernel-trace only -- MI355X_MICROARCH.md "HBM", "rocprofv3 PMC slots").  Everything lands in gpurun_out/<tag>/.
# usage: scripts/gpu_round.sh <tag> [bench flags...]
set -u
TAG=${1:-run}; shift || true
BFLAGS="$*"
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > "$OUT/pytest_gpu.log"
tail -3 "$OUT/pytest_gpu.log"
( timeout 600 python bench.py $BFLAGS 2> "$OUT/bench.err" | tail -1 ) > "$OUT/bench_line.json"
cat "$OUT/bench_line.json"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o stats -- python "$GRAFT_REPO_ROOT/bench.py" $BFLAGS --no-cpu-baseline > "$OUT/bench_prof_line.json" 2> "$OUT/prof.err"
# counters only for our kernels (the torch signal generator launches ~60k tiny kernels); FETCH_SIZE = 3 TCC counters did not finish
# on this pool in r1_v3, so the read side is taken from TCC_EA0_RDREQ_sum (1 counter) first and FETCH_SIZE is tried last
KRE='k_oqpsk|k_msk|k_coarse|k_burst|k_hilbert|k_trident|k_hist|k_aerol|k_viterbi'
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" $BFLAGS --no-cpu-baseline --steps 4 --warmup 2 > "$OUT/pmc_write.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d "$OUT/pmc_rdreq" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" $BFLAGS --no-cpu-baseline --steps 4 --warmup 2 > "$OUT/pmc_rdreq.log" 2>&1
timeout 240 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" $BFLAGS --no-cpu-baseline --steps 4 --warmup 2 > "$OUT/pmc_fetch.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt" | tail -30
find "$OUT" -name "*.csv" -size +8M -delete
du -sh "$OUT"
