"""bench_state.py -- what bench.py records AROUND its timed region so that a line explains itself (the one driver-timed
number fell 20 % on device code identical to HEAD and nothing in the line could tell a clock- or power-limited box from a regression).

  GpuStateSampler  a side thread reading the GPU's engine clock (per XCD), socket power, hot-spot temperature and the firmware's throttle
                   residency counters every ~50 ms through the amdsmi Python binding that ships with ROCm (/opt/rocm/share/amd_smi; one
                   amdsmi_get_gpu_metrics_info call, ~25 ms), sysfs hwmon (freq1_input, power1_input, power1_cap, temp2_input) if that is missing.
  Calib            the two fixed calibration kernels of jaero_amd/csrc/calib.hip (libjaero_calib.so), timed with HIP events on the bench's stream:
                   dependent-free fp64 FMA on every SIMD -> fp64_tflops, a 16-byte-per-lane copy of 2 x 4 GB -> hbm_gbs.
  step_stats       min / median / max of the per-step times (HIP events between the steps).
Measurement infrastructure only: nothing here is on the demodulator path."""
from __future__ import annotations

import ctypes as C
import glob
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
FP64_PEAK_TFLOPS = 78.6      # MI355X_MICROARCH.md: fp64 vector peak (256 CUs x 4 SIMDs x 16 lanes x 2 flops x 2.4 GHz)
HBM_PEAK_GBS = 8000.0


def _amdsmi():
    try:
        sys.path.insert(0, "/opt/rocm/share/amd_smi")
        import amdsmi  # noqa: F401

        return amdsmi
    except Exception:
        return None
    finally:
        if sys.path and sys.path[0] == "/opt/rocm/share/amd_smi":
            sys.path.pop(0)


def _num(v):
    return float(v) if isinstance(v, (int, float)) and not isinstance(v, bool) else None


class GpuStateSampler:
    """start() ... mark('timed') ... stop() -> summary(); every sample is (t, window name, sclk MHz mean over XCDs, sclk MHz min over XCDs, W, deg C)."""

    def __init__(self, pci_bus_id: str | None = None, period_s: float = 0.05):
        self.period = period_s
        self.samples = []
        self.window = "before"
        self._stop = threading.Event()
        self._thr = None
        self.source = None
        self.power_cap_w = None
        self._acc0 = self._acc1 = None
        self._h = None
        self._hwmon = None
        self.error = None
        smi = _amdsmi()
        if smi is not None:
            try:
                smi.amdsmi_init()
                hs = smi.amdsmi_get_processor_handles()
                h = hs[0]
                if pci_bus_id:
                    for x in hs:
                        try:
                            if smi.amdsmi_get_gpu_device_bdf(x).lower().endswith(pci_bus_id.lower()[-10:]):
                                h = x
                        except Exception:
                            pass
                self._smi, self._h, self.source = smi, h, "amdsmi"
                try:
                    self.power_cap_w = smi.amdsmi_get_power_cap_info(h)["power_cap"] / 1e6
                except Exception:
                    pass
            except Exception as e:
                self.error = f"amdsmi: {type(e).__name__}: {e}"[:200]
        if self._h is None:
            cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
            if pci_bus_id:
                m = [c for c in cands if pci_bus_id.lower() in os.path.realpath(c).lower()]
                cands = m or cands
            for c in cands:
                if os.path.exists(os.path.join(c, "freq1_input")):
                    self._hwmon, self.source = c, "sysfs hwmon"
                    try:
                        self.power_cap_w = int(open(os.path.join(c, "power1_cap")).read()) / 1e6
                    except Exception:
                        pass
                    break

    def _read(self):
        if self._h is not None:
            m = self._smi.amdsmi_get_gpu_metrics_info(self._h)
            clks = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float))] or [m.get("current_gfxclk")]
            clks = [float(c) for c in clks if isinstance(c, (int, float))]
            acc = {k: _num(m.get(k)) for k in ("accumulation_counter", "ppt_residency_acc", "prochot_residency_acc", "socket_thm_residency_acc",
                                                "vr_thm_residency_acc", "hbm_thm_residency_acc")}
            return (sum(clks) / len(clks) if clks else None, min(clks) if clks else None, _num(m.get("current_socket_power")),
                    _num(m.get("temperature_hotspot")), acc)
        if self._hwmon is not None:
            rd = lambda f: int(open(os.path.join(self._hwmon, f)).read())
            clk = rd("freq1_input") / 1e6
            return clk, clk, rd("power1_input") / 1e6, rd("temp2_input") / 1e3, None
        return None

    def _loop(self):
        while not self._stop.is_set():
            t = time.perf_counter()
            try:
                r = self._read()
                if r is not None:
                    self.samples.append((t, self.window) + r[:4])
                    if r[4] is not None and self.window == "timed":
                        if self._acc0 is None:
                            self._acc0 = r[4]
                        self._acc1 = r[4]
            except Exception as e:  # a sensor that stops answering must not take the bench with it
                self.error = f"{type(e).__name__}: {e}"[:200]
            self._stop.wait(max(0.0, self.period - (time.perf_counter() - t)))

    def start(self, window="sustain"):
        if self.source is None:
            return self
        self.window = window
        self._thr = threading.Thread(target=self._loop, daemon=True)
        self._thr.start()
        return self

    def mark(self, window):
        self.window = window

    def stop(self):
        self._stop.set()
        if self._thr is not None:
            self._thr.join(timeout=2.0)

    def _stats(self, window):
        s = [x for x in self.samples if x[1] == window and x[2] is not None]
        if not s:
            return None
        clk, clkmin = [x[2] for x in s], [x[3] for x in s]
        pw = [x[4] for x in s if x[4] is not None]
        tp = [x[5] for x in s if x[5] is not None]
        return {"n": len(s), "sclk_mhz_mean": round(sum(clk) / len(clk), 1), "sclk_mhz_min": round(min(clkmin), 1),
                "power_w_mean": round(sum(pw) / len(pw), 1) if pw else None, "power_w_max": round(max(pw), 1) if pw else None,
                "temp_hotspot_c_max": round(max(tp), 1) if tp else None}

    def summary(self):
        if self.source is None:
            return {"source": None, "error": self.error or "no amdsmi binding and no readable hwmon node"}
        out = {"source": self.source, "period_ms": int(self.period * 1e3), "power_cap_w": self.power_cap_w}
        t = self._stats("timed")
        if t:
            out.update(t)
        for w in ("sustain", "calib"):
            st = self._stats(w)
            if st:
                out[w] = {k: st[k] for k in ("n", "sclk_mhz_mean", "sclk_mhz_min", "power_w_mean", "power_w_max")}
        # throttle residency over the timed window: the firmware accumulates, per cause, the time the cause limited the clocks
        thr = None
        if self._acc0 and self._acc1 and self._acc0 is not self._acc1:
            d = lambda k: (self._acc1[k] - self._acc0[k]) if (self._acc0.get(k) is not None and self._acc1.get(k) is not None) else None
            tot = d("accumulation_counter")
            if tot and tot > 0:
                thr = {k.replace("_residency_acc", "_frac"): round(d(k) / tot, 4) for k in ("ppt_residency_acc", "prochot_residency_acc",
                                                                                               "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc") if d(k) is not None}
                thr = {"ppt_frac": thr.get("ppt_frac", 0.0), **{k: v for k, v in thr.items() if v > 0 and k != "ppt_frac"}}  # ppt = the socket power limit
        out["throttle_residency"] = thr
        cap = self.power_cap_w
        mean_clk = (t or {}).get("sclk_mhz_mean")
        out["throttled"] = bool((thr and any(v > 0.02 for v in thr.values())) or (mean_clk is not None and mean_clk < 0.93 * 2400.0))
        out["throttled_rule"] = "a throttle residency > 2 % of the timed window, or mean sclk < 93 % of 2400 MHz"
        if t and t.get("power_w_max") and cap:
            out["power_frac_of_cap_max"] = round(t["power_w_max"] / cap, 3)
        if self.error:
            out["error"] = self.error
        return out


class Calib:
    """ctypes over libjaero_calib.so; run() -> {"fp64_tflops", "hbm_gbs", ...}"""

    def __init__(self, device: int, copy_bytes: int = 4 << 30):
        path = os.path.join(ROOT, "jaero_amd", "libjaero_calib.so")
        self.lib = C.CDLL(path)  # raises if the library was not built (build() makes it)
        self.lib.jaero_calib_create.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
        self.lib.jaero_calib_fp64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        self.lib.jaero_calib_hbm_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        self.lib.jaero_calib_destroy.argtypes = [C.c_void_p]
        self.h = C.c_void_p()
        rc = self.lib.jaero_calib_create(device, copy_bytes, C.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"jaero_calib_create failed ({rc})")
        self.fp64_iters = 48000  # ~10 ms at the nominal clock: 1024 SIMDs x 2 wavefronts x 64 lanes x 48000 x 64 fma = 4.0e11 fma

    def run(self, stream: int, copies: int = 4):
        ms, fl, by = C.c_double(), C.c_double(), C.c_double()
        rc = self.lib.jaero_calib_fp64(self.h, C.c_void_p(stream), self.fp64_iters, C.byref(ms), C.byref(fl))
        if rc != 0:
            raise RuntimeError(f"jaero_calib_fp64 failed ({rc})")
        fp_ms, tfl = ms.value, fl.value / (ms.value * 1e-3) / 1e12
        rc = self.lib.jaero_calib_hbm_copy(self.h, C.c_void_p(stream), copies, C.byref(ms), C.byref(by))
        if rc != 0:
            raise RuntimeError(f"jaero_calib_hbm_copy failed ({rc})")
        return {"fp64_tflops": round(tfl, 2), "fp64_ms": round(fp_ms, 3), "hbm_gbs": round(by.value / (ms.value * 1e-3) / 1e9, 1), "hbm_copy_ms": round(ms.value, 3)}

    def close(self):
        if self.h:
            self.lib.jaero_calib_destroy(self.h)
            self.h = C.c_void_p()


def step_stats(ms_list):
    if not ms_list:
        return None
    s = sorted(ms_list)
    return {"min": round(s[0], 4), "p50": round(s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2]), 4), "max": round(s[-1], 4), "n": len(s)}
