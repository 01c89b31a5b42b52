#!/bin/bash
# what can an ordinary user read about clocks / power on the GPU box?  (round 6: bench.py's gpu_state sampler is built on the answer)
id; nproc
for d in /sys/class/drm/card*/device; do
  echo "== $d"; cat $d/vendor 2>/dev/null
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk gpu_busy_percent mem_busy_percent power_dpm_force_performance_level current_link_speed; do echo "-- $f"; cat $d/$f 2>&1 | head -12; done
  for h in $d/hwmon/hwmon*; do echo "-- $h"; ls $h; for f in $h/power1_average $h/power1_input $h/power1_cap $h/power1_cap_max $h/power1_cap_min $h/power1_cap_default $h/temp*_input $h/freq*_input $h/freq*_label; do echo "$f: $(cat $f 2>&1)"; done; done
  ls $d | tr '\n' ' '; echo
done
echo "== rocm-smi"; timeout 60 /opt/rocm/bin/rocm-smi --showclocks --showpower --showmaxpower --showtemp --showperflevel --json 2>&1 | head -40
echo "== amd-smi metric"; timeout 60 /opt/rocm/bin/amd-smi metric --json 2>&1 | head -150
echo "== amd-smi static limit"; timeout 60 /opt/rocm/bin/amd-smi static --limit --json 2>&1 | head -60
echo "== python amdsmi"; python - <<'PY'
import sys, time
sys.path.insert(0, "/opt/rocm/share/amd_smi")
try:
    import amdsmi
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    print("handles", len(hs))
    h = hs[0]
    t0 = time.perf_counter()
    for fn in ("amdsmi_get_gpu_metrics_info", "amdsmi_get_power_info", "amdsmi_get_power_cap_info", "amdsmi_get_clock_info", "amdsmi_get_violation_status"):
        try:
            f = getattr(amdsmi, fn)
            if fn == "amdsmi_get_clock_info":
                print(fn, f(h, amdsmi.AmdSmiClkType.GFX))
            else:
                print(fn, f(h))
        except Exception as e:
            print(fn, "ERR", repr(e)[:300])
    print("dt", time.perf_counter() - t0)
except Exception as e:
    print("amdsmi import/init failed:", repr(e)[:300])
PY
# (lowering the power cap is not possible on this pool: gpurun refuses any command that changes a machine-wide GPU setting, round 6)
