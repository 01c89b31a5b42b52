#!/bin/bash
# quick check of an Aero-L change: the Aero-L GPU tests, then the three Aero-L bench lines twice
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -x -q -k "aerol or qt or recording" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for wl in ${WLS:-aerol aerol_c aerol_burst}; do for i in 1 2; do
  extra=""; [ $wl = aerol ] && extra="--warmup 4"
  python bench.py --workload $wl $extra --no-cpu-baseline --no-state --sustain 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); oc=d['config'].get('oracle_check',{}); print('$wl', d['value'], d['ms_per_step'], d['config'].get('kernel_ms_per_step'), oc.get('rows_equal'), oc.get('events_equal'))"
done; done
