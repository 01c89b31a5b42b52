"""CPU: the pieces bench.py adds around its timed region in round 6 (bench_state.py, the other-workloads summary, the Aero-L issue roofline) must never take a
line with them: no sensor, no counter summary, a workload that printed nothing."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture()
def bench(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    import importlib

    import bench as B

    importlib.reload(B)
    B.ARGS = B.parse()
    return B


def test_step_stats_and_sampler_without_a_gpu():
    import bench_state as BS

    assert BS.step_stats([]) is None
    assert BS.step_stats([3.0, 1.0, 2.0]) == {"min": 1.0, "p50": 2.0, "max": 3.0, "n": 3}
    assert BS.step_stats([4.0, 1.0, 2.0, 3.0])["p50"] == 2.5
    s = BS.GpuStateSampler(pci_bus_id="0000:00:00.0", period_s=0.01).start("sustain")
    s.mark("timed")
    s.stop()
    out = s.summary()  # no GPU here: either no source at all, or a source that produced no sample -- a dictionary either way
    assert isinstance(out, dict) and "source" in out
    json.dumps(out)


def test_other_workload_summary(bench):
    line = {"value": 101.0, "unit": "Msamples/s", "ms_per_step": 2.5, "steps": 6,
            "roofline": {"bound": "hbm", "kernel_name": "k_x", "avg_launch_ms": 1.25, "frac": 0.2, "frac_of_calib_hbm": 0.33},
            "config": {"oracle_check": {"channels": [0, 1, 2], "hard_bits_equal": True, "max_soft_byte_diff": 0, "bits_compared": 10}},
            "gpu_state": {"sclk_mhz_mean": 2300.0, "power_w_mean": 900.0, "throttled": False}, "calib": {"fp64_tflops": 60.0, "hbm_gbs": 4800.0}, "step_ms": {"min": 1, "p50": 2, "max": 3, "n": 6}}
    s = bench.summarise_workload(line)
    assert s["value"] == 101.0 and s["dominant_kernel"] == "k_x" and s["kernel_ms"] == 1.25 and s["oracle_check"] == {"ok": True, "channels": 3, "max_soft_byte_diff": 0, "bits_compared": 10}
    assert s["sclk_mhz"] == 2300.0 and s["frac_of_calib_hbm"] == 0.33
    # a line without an oracle check is reported as NOT checked, never as ok
    assert bench.summarise_workload({"value": 1.0, "config": {}})["oracle_check"]["ok"] is False
    assert bench.summarise_workload({"value": 1.0, "config": {"oracle_check": {"channels": [0], "rows_equal": True, "events_equal": False}}})["oracle_check"]["ok"] is False
    assert len(json.dumps(s)) < 700  # eight of these go into the one line the driver records


def test_aerol_lines_are_priced_against_valu_issue(bench, monkeypatch):
    for wl in ("aerol", "aerol_burst", "aerol_c"):
        path = os.path.join(ROOT, "profiles", f"sq_summary_{wl}.json")
        assert os.path.exists(path), path
        bench.ARGS.workload = wl
        line = {"unit": "Msoftbits/s", "ms_per_step": 4.0, "config": {"channels_per_gpu": 65536},
                "roofline": {"bound": "hbm", "kernel_name": "k_viterbi_lanes", "achieved": 200.0, "frac": 0.03, "traffic": 1.0, "traffic_from": "x"}}
        bench.aerol_issue_roofline(line)
        r = line["roofline"]
        assert r["bound"] == "int_valu_issue" and r["unit"] == "wave-instructions/s" and r["hbm"]["frac"] == 0.03
        assert 0.1 < r["frac"] < 1.0 and abs(r["frac"] - r["floor_ms"] / 4.0) < 1e-3 and r["peak"] == 1024 * 2.4e9 / 4
    # no summary for the workload: the block says so instead of carrying an HBM fraction forward
    bench.ARGS.workload = "no_such_workload"
    line = {"unit": "Msoftbits/s", "ms_per_step": 4.0, "config": {"channels_per_gpu": 65536}, "roofline": {"bound": "hbm", "frac": 0.03}}
    bench.aerol_issue_roofline(line)
    assert line["roofline"]["bound"] == "int_valu_issue" and line["roofline"]["frac"] is None and "reason" in line["roofline"]


def test_emit_attaches_what_run_timed_recorded(bench, capsys, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.MEAS.clear()
    bench.MEAS.update({"gpu_state": {"source": "amdsmi", "power_w_mean": 1000.0, "throttled": True}, "calib": {"fp64_tflops": 60.0, "hbm_gbs": 5000.0, "fp64_frac_of_peak": 0.8},
                       "step_ms": {"min": 1.0, "p50": 1.0, "max": 1.1, "n": 20}, "sustain_steps": 130})
    line = {"metric": "m", "value": 10000.0, "unit": "Msamples/s", "n_gpus": 1, "config": {}, "roofline": {"bound": "hbm", "achieved": 2500.0, "unit": "GB/s", "frac": 0.3125},
            "roofline_fp64_issue": {"step": {"frac": 0.56}}}
    bench.emit(line)
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["gpu_state"]["value_per_watt"] == 10.0 and out["roofline"]["frac_of_calib_hbm"] == 0.5 and out["roofline_fp64_issue"]["step"]["frac_at_calib_clock"] == 0.7
    assert out["step_ms"]["n"] == 20 and out["config"]["sustain_steps"] == 130
    assert os.path.exists(os.path.join(str(tmp_path), "gpurun_out", "bench_details.json"))
    bench.MEAS.clear()
