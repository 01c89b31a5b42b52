"""Key fields of bench.py JSON lines (files given as arguments): value, ms per step, the dominant kernel's name / time / roofline fraction,
the post-clock oracle check."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
    except Exception as e:  # noqa: BLE001
        print(f, "unreadable:", e)
        continue
    r = d.get("roofline", {})
    oc = d.get("oracle_check")
    print(f"{f}: {d.get('value')} {d.get('unit')}  {d.get('ms_per_step')} ms/step  n_gpus {d.get('n_gpus')}  "
          f"kernel {r.get('kernel_name')} {r.get('kernel_ms_per_step', r.get('avg_launch_ms'))} ms frac {r.get('frac')} traffic {r.get('traffic')}  "
          f"oracle_check {json.dumps(oc)[:160] if oc else None}")
    pk = d.get("roofline_fp64_issue", {}).get("per_kernel")
    if pk:
        for k, v in pk.items():
            print(f"    {k}: {v.get('kernel')} ms {v.get('ms')} floor {v.get('floor_ms')} frac {v.get('frac')}")
