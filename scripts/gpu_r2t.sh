#!/bin/bash
# burst MSK kernel variants: parity tests, then the burst_msk workload
set -u
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests -m gpu -q -k "burst_msk or burstmsk or recording or (burst and msk) or adaptor" --tb=short 2>&1 | grep -E "^E   |passed|failed|^FAILED" | cut -c1-250
timeout 400 python bench.py --workload burst_msk --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['kernel_ms_total'])"
