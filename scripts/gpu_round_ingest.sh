#!/bin/bash
# Batched ingest (row f3) on the GPU box: its parity tests and the host-inclusive rate.  usage: scripts/gpu_round_ingest.sh <tag>
set -u
TAG=${1:-ingest}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
( timeout 600 python -m pytest tests/test_gpu_ingest.py -q 2>&1 | tail -15 ) > "$OUT/pytest_gpu.log"; tail -15 "$OUT/pytest_gpu.log"
hipcc -O2 -Wno-unused-result -Iinclude scripts/ingest_rate.cpp -Ljaero_amd -l:libjaero_hip.so -Wl,-rpath,"$GRAFT_REPO_ROOT/jaero_amd" -o /tmp/ingest_rate 2> "$OUT/build.err"
for cfg in "16384 4096 12" "65536 4096 6" "16384 1024 24"; do
  timeout 300 /tmp/ingest_rate $cfg 2>> "$OUT/rate.err" | tee -a "$OUT/ingest_rate.jsonl"
done
