#!/usr/bin/env python3
"""bench.py -- throughput of the demodulator hot path on N MI355X GPUs of one node (driver contract in the task).

A "step" = one jaero_write of `--chunk` (4096) samples for every channel of the bank: one sample-loop kernel launch
plus one coarse-frequency kernel launch (the reference runs its 2^14-point estimate every 4096 samples), on synthetic
10.5 kbps OQPSK PCM that is already resident in HBM as interleaved frames.  Channels are sharded across ranks with
no collective on the data path (weak scaling: `--channels` per GPU); the aggregate is all ranks' samples divided by
the slowest rank's time.

Printed JSON (one line, rank 0): metric Msamples/s (real 48 kHz PCM samples consumed per second, all channels),
plus `roofline` for the dominant kernel (HIP-event timed inside libjaero_hip on the launch stream) and
`cpu_baseline` (the unmodified reference, or the C port when the reference binary cannot run, on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per input sample, continuous OQPSK 10.5k (fp64 state as in the reference)
ALG_BYTES_SAMPLE_KERNEL = 2.0 + 16.0 + 16.0 + 0.44 + 0.5      # PCM + AGC ring r/w + coarse ring write + soft out + state
ALG_BYTES_SAMPLE_KERNEL_EBNO = 32.0                           # optional EbNo rings (2 x read+write)
ALG_BYTES_COARSE_KERNEL = 128.0                               # (ring read 256 KiB + y r/w 256 KiB) / 4096 samples
ALG_BYTES_WHOLE_PATH = 163.0                                  # SURVEY.md 8(d) headline
# burst OQPSK (column "Burst OQPSK"): 187 B/sample = PCM 2 + AGC ring 16 + burst rings 168 + soft/state 0.94
ALG_BURST = {"hilbert": 2.0, "front": 16.0 + 168.0, "demod": 0.44 + 0.5, "trident": 0.0}
ALG_BURST_EBNO = 32.0                                         # optional EbNo rings, charged to the tracking kernel
ALG_BYTES_WHOLE_PATH_BURST = 187.0
HBM_PEAK_GBS = 8000.0                                         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s


def child_env():
    """Environment for the CPU-baseline child processes: a profiler wrapped around bench.py (rocprofv3) preloads its tool library into
    every child, and that library needs a newer libstdc++ than the Qt build the reference binary runs against."""
    env = dict(os.environ)
    for k in list(env):
        if k == "LD_PRELOAD" or k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS")):
            env.pop(k)
    env.setdefault("QT_QPA_PLATFORM", "offscreen")
    return env


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--channels", type=int, default=int(os.environ.get("JAERO_BENCH_CHANNELS", "65536")), help="channels per GPU")
    ap.add_argument("--chunk", type=int, default=4096)
    ap.add_argument("--ebno", type=int, default=1, help="run the EbNo meters (the reference always does)")
    ap.add_argument("--ebno-db", type=float, default=10.0)
    ap.add_argument("--idle-frac", type=float, default=0.0, help="aerol workload: fraction of channels that carry noise only (never lock)")
    ap.add_argument("--preroll", type=int, default=-1, help="continuous OQPSK workloads: untimed steps before the warm-up (default: enough for t >= 4 s at the first timed step)")
    ap.add_argument("--timing-phases", type=int, default=32, help="continuous OQPSK workloads: number of distinct symbol-clock phases drawn per channel (1 = all channels symbol-synchronous)")
    ap.add_argument("--check-channels", type=int, default=16, help="channels (spread over the bank) compared with the oracle on the same PCM after the timed region; 0 = none")
    ap.add_argument("--as-written", type=int, default=1, help="also time BASELINE configs[2] / configs[1] at their literal sizes (N = 1)")
    ap.add_argument("--workload", default="oqpsk", choices=["oqpsk", "oqpsk8400", "msk", "burst_oqpsk", "burst_msk", "aerol", "aerol_burst", "aerol_c"],
                    help="oqpsk = BASELINE configs[2] (continuous, the headline); msk = configs[1] shape (1200 bps MSK) scaled to a bank that fills the chip; burst_oqpsk = configs[3] (one burst per second per "
                         "channel); aerol = the 10.5 kbps P-channel bit pipeline behind the demodulator (SURVEY 8 row f1), one frame per step; "
                         "aerol_burst = the R/T channel packet search behind a burst demodulator (row f2), one burst per channel and step")
    ap.add_argument("--sustain", type=float, default=3.0, help="continuous OQPSK workloads: seconds of untimed back-to-back steps in front of the warm-up (resident like the timed "
                                                               "steps), so that clocks and power are in steady state when the clock starts; 0 = none")
    ap.add_argument("--no-state", action="store_true", help="do not sample clocks / power and do not run the calibration kernels around the timed region")
    ap.add_argument("--no-other-workloads", action="store_true", help="headline only: do not run the other workloads behind it (config.other_workloads)")
    ap.add_argument("--other-steps", type=int, default=6, help="timed steps of each of the other workloads")
    ap.add_argument("--fb", type=float, default=1200.0, help="msk workload: 1200 or 600 bps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-samples", type=int, default=500_000, help="samples per process for the CPU baseline leg (>= 10 s of signal)")
    return ap.parse_args()


def maybe_spawn():
    """`python bench.py --gpus N` without a launcher around it: start the N ranks here (one per GPU, torch.distributed.run on
    127.0.0.1) and exit with their status.  Under a launcher (WORLD_SIZE set) this is a no-op."""
    if ARGS.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ARGS.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def setup_ranks():
    """(rank, world, local rank, torch device, ranks_share_a_device).  world must be what --gpus says: no silent single-rank run."""
    import torch

    from jaero_amd import dist as jd

    rank, world, local = jd.init_from_env()
    if world != ARGS.gpus:
        raise SystemExit(f"bench.py: --gpus {ARGS.gpus} but WORLD_SIZE = {world} (launch with torch.distributed.run --nproc-per-node {ARGS.gpus}, or without a launcher)")
    shared = jd.ranks_share_a_device()
    di = jd.device_index(local)
    torch.cuda.set_device(di)
    return rank, world, local, torch.device("cuda", di), shared


MEAS = {}   # what run_timed recorded around the last timed region (gpu_state, calib, step_ms): attached to the line by emit()


def run_timed(step, W, K, world, dev, before_timed=None, sustain=0):
    """`sustain` + W untimed steps back to back, then exactly K steps between barrier + device synchronisation on both sides.  Returns (max over
    ranks of the wall time between the barriers, every rank's time for its own K steps).  Around the timed region (bench_state.py): a side
    thread samples engine clock / socket power / throttle residency, the two calibration kernels run immediately before and after it, and an
    event behind every step gives the per-step times -- all of it lands in MEAS."""
    import torch
    import torch.distributed as dist

    import bench_state as BS

    gpu = torch.cuda.is_available()
    sync = torch.cuda.synchronize if gpu else (lambda: None)  # (no GPU: the gloo plumbing test of tests/test_dist_gloo.py)
    MEAS.clear()
    sampler = calib = None
    cal = {}
    stream = torch.cuda.current_stream().cuda_stream if gpu else 0
    if gpu and not ARGS.no_state:
        try:
            pr = torch.cuda.get_device_properties(dev.index)
            bdf = ("%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)) if hasattr(pr, "pci_bus_id") else None
        except Exception:
            bdf = None
        sampler = BS.GpuStateSampler(pci_bus_id=bdf)
        try:
            calib = BS.Calib(dev.index)
            cal["cold"] = calib.run(stream)
        except Exception as e:
            cal["error"] = f"{type(e).__name__}: {e}"[:200]
            calib = None
        sampler.start("sustain")
    for i in range(sustain + W):
        step(i)
    sync()
    if before_timed is not None:
        before_timed()
    if calib is not None:
        sampler.mark("calib")
        cal["before"] = calib.run(stream)
    if world > 1:
        dist.barrier()
    sync()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)] if gpu else None
    if sampler is not None:
        sampler.mark("timed")
    t0 = time.perf_counter()
    if ev:
        ev[0].record()
    for i in range(sustain + W, sustain + W + K):
        step(i)
        if ev:
            ev[i - sustain - W + 1].record()
    sync()
    own = time.perf_counter() - t0      # this rank's own K steps (before it waits for the others)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if sampler is not None:
        sampler.mark("calib")
    if calib is not None:
        cal["after"] = calib.run(stream)
        calib.close()
    if sampler is not None:
        sampler.stop()
        MEAS["gpu_state"] = sampler.summary()
    if cal:
        ok = [cal[k] for k in ("before", "after") if k in cal]
        if ok:
            cal["fp64_tflops"] = round(sum(x["fp64_tflops"] for x in ok) / len(ok), 2)
            cal["hbm_gbs"] = round(sum(x["hbm_gbs"] for x in ok) / len(ok), 1)
            cal["fp64_frac_of_peak"] = round(cal["fp64_tflops"] / BS.FP64_PEAK_TFLOPS, 4)
            cal["what"] = "fp64 FMA on every SIMD ~10 ms; 16 B/lane copy 2 x 4 GB; HIP events right before / after the timed steps (jaero_amd/csrc/calib.hip)"
        MEAS["calib"] = cal
    if ev:
        MEAS["step_ms"] = BS.step_stats([ev[i].elapsed_time(ev[i + 1]) for i in range(K)])
    MEAS["sustain_steps"] = sustain
    dts = [own]
    if world > 1:
        c = MEAS.get("calib", {})
        g = MEAS.get("gpu_state", {})
        t = torch.tensor([dt, own, c.get("fp64_tflops") or 0.0, c.get("hbm_gbs") or 0.0, g.get("sclk_mhz_mean") or 0.0, g.get("power_w_mean") or 0.0],
                         dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        lst = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(lst, t)
        dt = max(float(x[0].item()) for x in lst)
        dts = [float(x[1].item()) for x in lst]
        MEAS["per_rank"] = {"fp64_tflops": [round(float(x[2].item()), 2) for x in lst], "hbm_gbs": [round(float(x[3].item()), 1) for x in lst],
                            "sclk_mhz_mean": [round(float(x[4].item()), 1) for x in lst], "power_w_mean": [round(float(x[5].item()), 1) for x in lst]}
    return dt, dts


def aerol_issue_roofline(line):
    """The Aero-L bit pipelines are integer-VALU work (unique-word correlation, deinterleave index arithmetic, add-compare-select): an HBM
    fraction says nothing about them.  Their roofline block is priced against VALU issue instead: every wavefront
    instruction holds its SIMD's issue port for 4 cycles, so a step cannot take less than (VALU wave-instructions of all its kernels) x 4 /
    (1024 SIMDs x 2.4 GHz).  Instruction counts: profiles/sq_summary_<workload>.json (rocprofv3 --pmc SQ_INSTS_VALU pass of this command,
    scripts/pmc_sq.sh); the HBM figures the line used to lead with stay under "hbm"."""
    old = line.get("roofline") or {}
    new = {"bound": "int_valu_issue", "kernel": old.get("kernel_name") or old.get("kernel"), "unit": "wave-instructions/s", "peak": SIMDS * CLOCK_HZ / 4.0,
           "achieved": None, "frac": None, "traffic": old.get("traffic"), "traffic_from": old.get("traffic_from"),
           "hbm": {k: old.get(k) for k in ("achieved", "frac", "alg_bytes_per_softbit", "avg_launch_ms", "softbits_per_launch", "whole_step") if k in old}}
    path = os.path.join(ROOT, "profiles", f"sq_summary_{ARGS.workload}.json")
    try:
        sq = json.load(open(path))
        steps = float(sq["steps_total"])
        scale = line["config"]["channels_per_gpu"] / float(sq.get("channels_per_gpu", 65536))
        per = {k: v["SQ_INSTS_VALU_sum"] / steps * scale for k, v in sq.items() if isinstance(v, dict) and "SQ_INSTS_VALU_sum" in v}
        tot = sum(per.values())
        floor = tot * 4.0 / (SIMDS * CLOCK_HZ) * 1e3
        ms = float(line["ms_per_step"])
        new.update({"kernel": "whole step: " + " + ".join(sorted(per, key=per.get, reverse=True)[:4]), "valu_insts_per_step": round(tot), "floor_ms": round(floor, 4), "ms": ms,
                    "achieved": round(tot / (ms * 1e-3), 1), "frac": round(floor / ms, 5),
                    "per_kernel_floor_ms": {k.split("(")[0][:40]: round(v * 4.0 / (SIMDS * CLOCK_HZ) * 1e3, 4) for k, v in per.items()},
                    "source": f"profiles/sq_summary_{ARGS.workload}.json@{sq.get('tag', 'untagged')} ({same_pass(sq).replace('--pmc passes', '--pmc SQ_INSTS_VALU pass')})"})
    except Exception as e:
        new["reason"] = f"no instruction counts for this workload ({type(e).__name__}: {e})"[:200]
    line["roofline"] = new


def emit(line):
    """Print the ONE JSON line, with what run_timed recorded around the timed region: gpu_state (clock, power, throttle residency), calib (the two
    calibration kernels before / after), step_ms (per-step spread), and the roofline fraction also relative to the calibrated rates of THIS box."""
    if "gpu_state" in MEAS:
        line["gpu_state"] = MEAS["gpu_state"]
    if "calib" in MEAS:
        line["calib"] = MEAS["calib"]
    if MEAS.get("step_ms"):
        line["step_ms"] = MEAS["step_ms"]
    pw = (MEAS.get("gpu_state") or {}).get("power_w_mean")
    if pw and line.get("value") and line.get("n_gpus") == 1:
        line["gpu_state"]["value_per_watt"] = round(line["value"] / pw, 3)  # the metric's unit per watt of socket power during the timed steps
    if "per_rank" in MEAS:
        line["config"]["per_rank_state"] = MEAS["per_rank"]
    if MEAS.get("sustain_steps"):
        line["config"]["sustain_steps"] = MEAS["sustain_steps"]
    c = MEAS.get("calib", {})
    r = line.get("roofline")
    if isinstance(r, dict) and r.get("achieved") and c.get("hbm_gbs") and r.get("unit") == "GB/s":
        r["frac_of_calib_hbm"] = round(r["achieved"] / c["hbm_gbs"], 5)
    iss = line.get("roofline_fp64_issue")
    if isinstance(iss, dict) and c.get("fp64_frac_of_peak") and isinstance(iss.get("step"), dict) and iss["step"].get("frac"):
        iss["step"]["frac_at_calib_clock"] = round(iss["step"]["frac"] / c["fp64_frac_of_peak"], 4)
    if line.get("unit") == "Msoftbits/s":
        aerol_issue_roofline(line)
    # the long form stays available beside the line (the driver's record keeps the tail of stdout: the line itself stays short)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_details.json"), "a") as f:
            f.write(json.dumps(line) + "\n")
    except OSError:
        pass
    print(json.dumps(line), flush=True)


def rank_fields(world, shared, dts, units_per_rank, scale=1e6):
    """config entries every multi-rank line carries: per-rank rate, whether ranks really had a GPU each."""
    return {"ranks": world, "per_rank_rate": [round(units_per_rank / t / scale, 2) for t in dts],
            "ranks_share_a_device": bool(shared),
            "control_plane": "none (single rank)" if world == 1 else ("gloo (ranks share a device: RCCL refuses that; throughput figure NOT a scaling point)" if shared else "nccl (RCCL)")}


def finish(world):
    import torch.distributed as dist

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


SIMDS, CLOCK_HZ = 1024, 2.4e9   # MI355X: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock


def issue_roofline(kernel_ms_per_step: dict, names: dict, nch: int):
    """The roof that binds these kernels: every VALU instruction of a wavefront holds its SIMD's issue port for 4
    cycles (fp64; 32-bit operations finish sooner, so the floor is generous), so a launch cannot take less than
    valu_insts x 4 / (1024 SIMDs x 2.4 GHz).  Instruction counts come from the SQ counter summary of the same workload
    (profiles/sq_summary*.json, scripts/pmc_sq.sh) and are only used when they were taken on the kernel that ran here."""
    path = os.path.join(ROOT, "profiles", "sq_summary.json")
    out = {"cycles_per_inst": 4, "simds": SIMDS, "clock_hz": CLOCK_HZ, "source": None, "per_kernel": {}, "step": None}
    if not os.path.exists(path):
        out["reason"] = "profiles/sq_summary.json missing (run scripts/gpu_round.sh)"
        return out
    try:
        sq = json.load(open(path))
    except Exception as e:
        out["reason"] = f"unreadable: {e}"
        return out
    out["source"] = f"profiles/sq_summary.json@{sq.get('tag', 'untagged')} ({same_pass(sq).replace('--pmc passes', '--pmc SQ_INSTS_VALU pass')})"
    scale = nch / float(sq.get("channels_per_gpu", nch))
    tot_floor = tot_ms = 0.0
    for cls, ms in kernel_ms_per_step.items():
        want = names.get(cls, "")
        ent = next((v for k, v in sq.items() if isinstance(v, dict) and want and want.rstrip("<") in k), None)
        if ent is None or "SQ_INSTS_VALU" not in ent:
            out["per_kernel"][cls] = {"valu_insts_per_step": None, "reason": f"no counters for the kernel that ran ({want!r})"}
            continue
        insts = ent["SQ_INSTS_VALU"] * scale * float(ent.get("full_launches_per_step", 1))
        floor = insts * 4.0 / (SIMDS * CLOCK_HZ) * 1e3
        out["per_kernel"][cls] = {"kernel": want, "valu_insts_per_step": round(insts), "floor_ms": round(floor, 3), "ms": round(ms, 3),
                                  "frac": round(floor / ms, 4) if ms else None, "wait_any_frac_of_wave_cycles": (round(ent["SQ_WAIT_ANY"] / ent["SQ_WAVE_CYCLES"], 3)
                                  if "SQ_WAIT_ANY" in ent and ent.get("SQ_WAVE_CYCLES") else None)}
        tot_floor += floor
        tot_ms += ms
    if tot_ms:
        out["step"] = {"floor_ms": round(tot_floor, 3), "ms": round(tot_ms, 3), "frac": round(tot_floor / tot_ms, 4)}
    return out


def measured_traffic(pmc_file: str, cls: str, kernel_that_ran: str, nch: int):
    """(traffic bytes per launch or None, where it came from / why it is null).  The counter summary is a committed file: it is used only
    if it names the kernel that ran here."""
    path = os.path.join(ROOT, "profiles", pmc_file)
    if not os.path.exists(path):
        return None, f"profiles/{pmc_file} missing"
    try:
        pj = json.load(open(path))
        ent = pj.get(cls)
        if not ent or ent.get("hbm_bytes_per_launch") is None:
            return None, f"profiles/{pmc_file} has no entry for {cls}"
        if not kernel_that_ran or kernel_that_ran.rstrip("<") not in str(ent.get("kernel", "")):
            return None, (f"STALE: profiles/{pmc_file}@{pj.get('tag', 'untagged')} was taken on {ent.get('kernel')!r}, this run launched {kernel_that_ran!r}: "
                          f"redo the --pmc passes (scripts/gpu_evidence.sh)")
        t = ent["hbm_bytes_per_launch"] * nch / float(pj.get("channels_per_gpu", nch))
        return t, f"profiles/{pmc_file}@{pj.get('tag', 'untagged')} ({same_pass(pj)}; kernel name checked)"
    except Exception as e:
        return None, f"profiles/{pmc_file} unreadable: {e}"


def same_pass(pj) -> str:
    """Where a counter summary comes from relative to this run: the same scripts/gpu_evidence.sh pass (it exports its tag) or an earlier one."""
    tag = os.environ.get("JAERO_EVIDENCE_TAG")
    if tag and str(pj.get("tag", "")).startswith(tag + "_"):
        return "rocprofv3 --pmc passes of this command taken minutes earlier in the same scripts/gpu_evidence.sh pass on the same box, not this process"
    return "rocprofv3 --pmc passes of this command from an earlier pass, not this run"


def burst_traffic(pmc_file: str, kernel_that_ran: str, nch: int):
    """As measured_traffic, for the burst workloads' summaries (entries keyed by kernel name): HBM bytes per launch of the dominant kernel,
    mean over the launches that did work (>= 10 % of the largest; a burst kernel's launches differ with the bursts in flight)."""
    path = os.path.join(ROOT, "profiles", pmc_file)
    if not os.path.exists(path):
        return None, f"profiles/{pmc_file} missing: no --pmc pass of this workload committed"
    try:
        pj = json.load(open(path))
        for k, ent in pj.items():
            if isinstance(ent, dict) and kernel_that_ran and kernel_that_ran.rstrip("<") in k and ent.get("hbm_bytes_per_launch") is not None:
                t = ent["hbm_bytes_per_launch"] * nch / float(pj.get("channels_per_gpu", nch))
                return t, f"profiles/{pmc_file}@{pj.get('tag', 'untagged')} ({same_pass(pj)}; kernel name checked)"
        return None, f"STALE: profiles/{pmc_file}@{pj.get('tag', 'untagged')} has no entry for {kernel_that_ran!r}: redo the --pmc passes (scripts/gpu_evidence.sh)"
    except Exception as e:
        return None, f"profiles/{pmc_file} unreadable: {e}"


def aerol_kernel_name(cls: str, nch: int, mode: str) -> str:
    """The kernel that does the work of an Aero-L kernel class at this bank size (the Viterbi picks its layout by size: one block per lane from
    16 384 blocks on, one per wavefront below: jaero_hip.hip viterbi_use_lanes)."""
    vit = "k_viterbi_lanes" if nch >= 16384 else "k_viterbi"
    if mode == "c":
        return {"bits": "k_aerolc_bits", "viterbi": vit, "post": "k_aerolc_post"}[cls]
    if mode == "b":
        return {"bits": "k_aerolb_bits", "viterbi": vit, "post": "k_aerolb_post"}[cls]
    return {"bits": "k_aerol_bits", "viterbi": vit, "post": "k_aerol_post_packed" if nch >= 16384 else "k_aerol_post"}[cls]


def aerol_traffic(pmc_file: str, kernel: str, nch: int):
    """HBM bytes per working launch of `kernel` from a committed counter summary (entries keyed by kernel name), or (None, why)."""
    return burst_traffic(pmc_file, kernel, nch)


def cpu_baseline(chunk: int):
    """The reference's own CPU path timed on this box's host cores: one process per core (function-local statics
    make instances unshareable), each demodulating `n` samples of the same kind of synthetic signal."""
    from jaero_amd import signalgen as G
    from oracle import oracle as O  # cpu_baseline leg only

    ncores = os.cpu_count() or 1
    n = ARGS.cpu_samples
    burst = ARGS.workload in ("burst_oqpsk", "burst_msk")
    bmsk = ARGS.workload == "burst_msk"
    refkind = "burstmsk" if bmsk else ("burstoqpsk" if burst else "oqpsk")
    if bmsk:
        pcm, _ = G.burst_msk(n, burst_starts=list(range(20000, n - 60000, 72000)), ndata=1000, fb=1200.0, fc=1010.0, ebno_db=18.0, seed=G.SEED_BASE + 77)
    elif burst:
        pcm, _ = G.burst_oqpsk(n, burst_starts=list(range(20000, n - 40000, 48000)), ndata_sym=3040, fc=8037.5, ebno_db=15.0, seed=G.SEED_BASE + 77)
    else:
        pcm, _ = G.oqpsk(n, fc=8037.5, ebno_db=ARGS.ebno_db, seed=G.SEED_BASE + 77)
    kind = "port"
    use_ref = O.have_ref()
    if use_ref:
        try:
            subprocess.check_output([O.REF_BIN, "fft", "/dev/null", "/dev/null", "n=1"], stderr=subprocess.STDOUT, env=child_env())
        except Exception:
            use_ref = False
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "in.s16")
        pcm.tofile(path)
        t0 = time.time()
        if use_ref:
            kind = "reference"
            extra = ["fb=1200", "lockingbw=1800", "freq_center=1000"] if bmsk else []
            procs = [subprocess.Popen([O.REF_BIN, "time", refkind, path, f"chunk={chunk}"] + extra, stdout=subprocess.PIPE, env=child_env()) for _ in range(ncores)]
            outs = [p.communicate()[0] for p in procs]
            inner = [float(o.split()[0]) for o in outs]
        else:
            mk = "O.BurstDemod(O.burst_msk_settings())" if bmsk else ("O.BurstDemod(O.burst_oqpsk_settings())" if burst else "O.Demod(O.oqpsk_settings())")
            code = ("import sys,time,numpy as np; sys.path.insert(0,%r); from oracle import oracle as O; "
                    "x=np.fromfile(%r,dtype=np.int16); d=%s; t=time.time(); "
                    "[d.write(x[s:s+%d]) for s in range(0,len(x),%d)]; print(time.time()-t)") % (ROOT, path, mk, chunk, chunk)
            procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, env=child_env()) for _ in range(ncores)]
            outs = [p.communicate()[0] for p in procs]
            inner = [float(o.split()[0]) for o in outs]
        wall = time.time() - t0
    # aggregate over cores: every core did n samples in `inner[i]` seconds of writeData time, concurrently
    value = sum(n / t for t in inner) / 1e6
    return {"value": round(value, 3), "unit": "Msamples/s", "cores": ncores, "kind": kind,
            "sample": f"{n} samples of 48 kHz {'1200 bps burst MSK' if bmsk else ('10.5k burst OQPSK' if burst else '10.5k OQPSK')} per core, {chunk}-sample writes, cpuReduce=false, "
                      f"one process per core ({wall:.1f} s wall)",
            "per_core_msps": round(value / ncores, 3),
            "note": "FFT inside the reference build is the JFFT stand-in (oracle/ref/shim/jfft.h), not JFFT"}


def ber_check(bank, bits, nch_check: int, tail: int = 3000):
    """Hard decisions of the last `tail` symbols of a few channels against the transmitted bits (4 ambiguity states)."""
    worst, locked = 0.0, 0
    for c in range(nch_check):
        soft = bank.read_softbits(c, cap=1 << 22)
        st = bank.read_status(c)
        locked += int(st.signal)
        if len(soft) < 2 * tail + 64:
            worst = 1.0
            continue
        hard = (soft >= 128).astype(np.uint8)
        im, re = hard[0::2][-tail:], hard[1::2][-tail:]
        b = bits[c].cpu().numpy()
        arms = (b[0::2], b[1::2])
        # both output streams must decode an arm: score the worse of (im best, re best)
        def arm_ber(stream):
            bb = 1.0
            for arm in arms:
                for lag in range(max(0, len(arm) - tail - 600), len(arm) - tail):
                    e = float(np.mean(arm[lag:lag + tail] != stream))
                    bb = min(bb, e, 1.0 - e)
            return bb
        worst = max(worst, arm_ber(im), arm_ber(re))
    return worst, locked


def burst_line(bank, rank, world, nch, chunk, K, W, dt, value, msk=False, extra=None):
    """JSON line for the burst OQPSK workload (BASELINE configs[3]); kernel classes: tracking chain, trident check, history
    push, Hilbert FIR, front end (jaero_profile_read which = 0..4)."""
    from jaero_amd import capi

    names = ["demod", "trident", "hist_push", "hilbert", "front"]
    ms, nl = {}, {}
    for w, nm in enumerate(names):
        ms[nm], nl[nm] = bank.profile_read(w)
    acc = 0
    for c in range(min(8, nch)):
        ev = bank.read_events(c)
        acc += int(np.sum((ev[:, 1] == capi.EV_SIGNAL) & (ev[:, 2] > 0)))
    if rank != 0:
        return
    dom = max(("demod", "front", "hilbert", "trident"), key=lambda k: ms[k])
    per_sample = ALG_BURST[dom] + (ALG_BURST_EBNO if dom == "demod" else 0.0)
    launches = max(nl[dom], 1)
    avg_ms = ms[dom] / launches
    units = K * chunk * nch / launches
    achieved = per_sample * units / (avg_ms * 1e-3) / 1e9
    traffic, traffic_from = burst_traffic("pmc_summary_burst_msk.json" if msk else "pmc_summary_burst_oqpsk.json", bank.profile_kernel(names.index(dom)), nch)
    line = {
        "metric": ("Msamples/s of real 48 kHz PCM through the 1200 bps burst MSK demodulator hot path" if msk else
                   "Msamples/s of real 48 kHz PCM through the 10.5 kbps burst OQPSK demodulator hot path"),
        "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"{nch}-channel-per-GPU synthetic 48 kHz 1200 bps burst MSK (R/T-channel style): one burst per 1.5 s per channel "
                                f"(120 bit periods carrier + 100 preamble + 1000 data bits), 32 distinct streams with their bursts at different "
                                f"offsets replicated over the channels, noise between bursts, Eb/N0 18 dB, {chunk}-sample writes" if msk else
                                f"{nch}-channel-per-GPU synthetic 48 kHz 10.5 kbps burst OQPSK (BASELINE configs[3] shape): one burst per "
                                f"second per channel (128 symbols carrier + 128 symbols preamble + 3040 symbols data) at a random offset, noise "
                                f"between bursts, Eb/N0 15 dB, {chunk}-sample writes"),
                   "channels_per_gpu": nch, "total_channels": nch * world, "chunk": chunk,
                   "realtime_channel_equivalents": int(value / 0.048),
                   "bursts_accepted_in_first_channels": acc, "channels_checked": min(8, nch),
                   "whole_path_hbm_frac_at_187B_per_sample": round(value * 1e6 * ALG_BYTES_WHOLE_PATH_BURST / 1e9 / (HBM_PEAK_GBS * world), 5),
                   "kernel_ms_total": {k: round(v, 3) for k, v in ms.items()}, "kernel_launches": nl},
        "roofline": {"bound": "hbm", "kernel": dom, "kernel_name": bank.profile_kernel(names.index(dom)), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_from": traffic_from,
                     "alg_bytes_per_sample": per_sample, "samples_per_launch": units, "avg_launch_ms": round(avg_ms, 4)},
    }
    tr, tr_from = measured_traffic("pmc_summary_burst_msk.json" if msk else "pmc_summary_burst_oqpsk.json", dom, bank.profile_kernel(names.index(dom)), nch)
    if tr is not None or "STALE" in tr_from:
        line["roofline"]["traffic"], line["roofline"]["traffic_from"] = tr, tr_from
    if extra:
        line["config"].update(extra)
    if world == 1 and not ARGS.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(chunk)
        except Exception as e:
            line["cpu_baseline"] = {"value": None, "error": str(e)}
    emit(line)



def aerol_oracle_check(kind: str, bank, nch: int, stream_of, nrows_cap: int, first: int = 4):
    """The three Aero-L workloads' post-clock check: sampled channels of the timed bank (wave edges, neighbouring wavefronts, the end)
    against the oracle fed the SAME soft bits -- signal-unit rows / packets / voice rows and event rows EXACTLY equal.  Reads consume, so
    every channel is read once here; returns (oracle_check dict, CRC-clean units or packets in the first `first` channels)."""
    from oracle import oracle as O  # checker only, after the clock has stopped

    O.build()
    check = spread_channels(nch, ARGS.check_channels) if ARGS.check_channels > 0 else []
    oc = {"channels": [int(c) for c in check], "rows_equal": True, "events_equal": True, "rows_compared": 0, "events_compared": 0, "crc_clean_rows": 0}
    good, cache = 0, {}
    for c in sorted(set(check) | set(range(min(first, nch)))):
        if kind == "burst":
            got = {"rows": bank.read_packets(c, nrows_cap), "events": bank.read_events(c, 4096)}
            good += len(got["rows"]) if c < first else 0
        else:
            got = {"rows": bank.read_sus(c, nrows_cap), "events": bank.read_events(c, 4096)}
            if kind == "c":
                got["voice"] = bank.read_voice(c, nrows_cap)
            good += int(got["rows"][:, 14].sum()) if c < first else 0
        if c not in check:
            continue
        x = stream_of(c)
        key = x.tobytes()
        if key not in cache:
            if kind == "p":
                o = O.run_aerol(10500, x, 1 << 20)
                cache[key] = {"rows": o["sus"], "events": o["events"]}
            elif kind == "burst":
                o = O.run_aerol_burst(10500, x)
                cache[key] = {"rows": O.packets_from_rows(o["packets"]), "events": o["events"]}
            else:
                a = O.AeroL(8400)
                for s_ in range(0, len(x), 32):
                    a.write(x[s_:s_ + 32])
                fn, voice = a.take_voice()
                cache[key] = {"voice": (fn, voice), "rows": a.take_sus(), "events": a.take_events()}
        want = cache[key]
        same_rows = (got["rows"] == want["rows"]) if kind == "burst" else bool(np.array_equal(got["rows"], want["rows"]))
        if kind == "c":
            same_rows = same_rows and bool(np.array_equal(got["voice"][0], want["voice"][0])) and bool(np.array_equal(got["voice"][1], want["voice"][1]))
            oc["voice_rows_compared"] = oc.get("voice_rows_compared", 0) + len(want["voice"][0])
        oc["rows_equal"] &= bool(same_rows)
        oc["events_equal"] &= bool(np.array_equal(got["events"], want["events"]))
        oc["rows_compared"] += len(want["rows"])
        oc["events_compared"] += len(want["events"])
        if kind != "burst":
            oc["crc_clean_rows"] += int(np.asarray(want["rows"])[:, 14].sum()) if len(want["rows"]) else 0
    return oc, good


def aerol_check_or_exit(line, oc, what: str):
    line["config"]["oracle_check"] = oc
    if oc["channels"] and not (oc["rows_equal"] and oc["events_equal"] and oc["rows_compared"] > 0):
        emit(line)
        raise SystemExit(f"bench.py: {what} of a sampled channel differ from the oracle's on the same soft bits")


def aerol_bench():
    """Aero-L bit pipeline: a step = one 0.5 s P-channel frame (5250 soft bits) for every channel, soft bits resident in HBM.
    Metric: soft bits per second (all channels).  Algorithmic bytes per soft bit: 2 (int16 in) + 1 (deinterleaved block write) + 1
    (Viterbi read) + 0.5 + 0.5 (decoded bit write/read) + 1 (delay line r/w per decoded bit = 0.5 per soft bit x 2) = 6;
    per kernel: k_aerol_bits 3, k_viterbi 1.5, k_aerol_post 1.5."""
    import torch
    import torch.distributed as dist

    from jaero_amd import aerol_frames as AF
    from jaero_amd import capi
    from jaero_amd import dist as jd
    from jaero_amd.demodulator import AeroLBank

    capi.lib()
    rank, world, local, dev, shared = setup_ranks()
    local = dev.index
    nch, K, W = ARGS.channels, ARGS.steps, ARGS.warmup
    fb, flen = 10500, 5250
    nuniq = 64
    streams = []
    prng = np.random.default_rng(77 + rank)
    for u in range(nuniq):
        # every channel of a wavefront at its own frame phase (a random-length prefix of noise), as unsynchronised satellites would be
        bits, _ = AF.p_channel_bits(AF.random_payloads(K + W + 1, fb, seed=900 + u + 1000 * rank), fb, invert_i=bool(u & 1), invert_q=bool(u & 2))
        pre = prng.integers(0, 2, size=int(prng.integers(0, flen)), dtype=np.uint8)
        streams.append(AF.to_soft(np.concatenate([pre, bits])[: (K + W) * flen], sigma=25.0, seed=u))
    host = np.stack(streams)  # [nuniq, (K+W)*5250]
    soft = torch.from_numpy(host).to(dev)
    idx = torch.arange(nch, device=dev) % nuniq
    if ARGS.idle_frac > 0:  # noise-only channels scattered over the bank (every wavefront gets some): stream index nuniq = noise
        noise = np.clip(np.round(128 + prng.normal(0.0, 40.0, size=host.shape[1])), 0, 255).astype(np.int16)
        host = np.concatenate([host, noise[None, :]])
        soft = torch.cat([soft, torch.from_numpy(noise[None, :]).to(dev)])
        g = torch.Generator(device="cpu").manual_seed(5)
        idle = (torch.rand(nch, generator=g) < ARGS.idle_frac).to(dev)
        idx = torch.where(idle, torch.full_like(idx, nuniq), idx)
    idx_host = idx.cpu().numpy()
    counts = torch.full((nch,), flen, dtype=torch.int32, device=dev)
    bank = AeroLBank(nch, fb, device=local, max_softbits_per_write=flen + 8, su_capacity=26 * (K + W) + 8)
    stream = torch.cuda.current_stream().cuda_stream
    pitch = (flen + 7) // 8 * 8  # 16-byte aligned rows, as a demodulator bank's soft-bit buffer has
    # every step's input resident in HBM before the clock starts (a demodulator bank would have written it in place)
    frames = torch.empty((K + W, nch, pitch), dtype=torch.int16, device=dev)
    for i in range(K + W):
        frames[i, :, :flen].copy_(soft[idx, i * flen:(i + 1) * flen])
    del soft

    def step(i):
        bank.write_device(frames[i].data_ptr(), counts.data_ptr(), pitch, flen, stream)

    dt, dts = run_timed(step, W, K, world, dev, before_timed=lambda: bank.profile_enable(True))
    names = ["bits", "viterbi", "post"]
    alg = {"bits": 3.0, "viterbi": 1.5, "post": 1.5}
    ms, nl = {}, {}
    for w, nm in enumerate(names):
        ms[nm], nl[nm] = bank.profile_read(w)
    oc, good = aerol_oracle_check("p", bank, nch, lambda c: host[idx_host[c]][: (K + W) * flen], 26 * (K + W) + 8)
    if rank == 0:
        value = float(K) * flen * nch * world / dt / 1e6
        dom = max(names, key=lambda k: ms[k])
        # every write launches each kernel class once per round (3 rounds for a 5250-bit write); only one round per step has a block
        # to decode, the others return at once -- the per-launch figures below are per WORKING launch = per step
        avg_ms = ms[dom] / K
        units = flen * nch
        achieved = alg[dom] * units / (avg_ms * 1e-3) / 1e9
        line = {
            "metric": "Msoftbits/s through the Aero-L P-channel bit pipeline (unique word, deinterleave, Viterbi, descramble, CRC)",
            "value": round(value, 2), "unit": "Msoftbits/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{nch}-channel-per-GPU 10.5 kbps P-channel frames (5250 soft bits = 0.5 s per step and channel), "
                                   f"{nuniq} distinct noisy frame streams at random frame phases replicated over the channels, arm inversions mixed",
                       "channels_per_gpu": nch, "total_channels": nch * world, "realtime_channel_equivalents": int(value * 1e6 / 10500),
                       "idle_channel_fraction": ARGS.idle_frac,
                       "crc_clean_units_in_first_channels": good, "channels_checked": min(4, nch),
                       "kernel_ms_per_step": {k: round(v / K, 4) for k, v in ms.items()}, "kernel_launches": nl},
            "roofline": {"bound": "hbm", "kernel": {"bits": "k_aerol_bits+k_aerol_bulk+k_aerol_deint", "viterbi": "k_viterbi_lanes", "post": "k_aerol_post"}[dom],
                         "kernel_name": aerol_kernel_name(dom, nch, "p"),
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": aerol_traffic("pmc_summary_aerol.json", aerol_kernel_name(dom, nch, "p"), nch)[0],
                         "traffic_from": aerol_traffic("pmc_summary_aerol.json", aerol_kernel_name(dom, nch, "p"), nch)[1], "alg_bytes_per_softbit": alg[dom],
                         "softbits_per_launch": units, "avg_launch_ms": round(avg_ms, 4),
                         "note": "the Viterbi is integer-VALU bound, not HBM bound: 225 wave instructions per trellis step for 64 blocks, "
                                 "one wavefront per SIMD issuing one instruction per ~2.2 ns (scripts/ubench/valu_rates.hip); "
                                 "the HBM figure is reported because the contract asks for it"},
        }
        line["config"].update(rank_fields(world, shared, dts, value * 1e6 * dt / world))
        aerol_check_or_exit(line, oc, "signal-unit rows / events")
        if world == 1 and not ARGS.no_cpu_baseline:
            from oracle import oracle as O  # cpu_baseline leg only
            x = np.tile(host[0], max(1, int(2_000_000 / host.shape[1]) + 1))[:2_000_000]
            ncores = os.cpu_count() or 1
            if O.have_ref():
                # the unmodified AeroL (oracle/_ref), one process per host core, each decoding the same 2 M soft bits in groups of 32
                with tempfile.TemporaryDirectory() as td:
                    path = os.path.join(td, "in.s16")
                    x.tofile(path)
                    t1 = time.time()
                    procs = [subprocess.Popen([O.REF_BIN, "aerol", path, os.path.join(td, f"o{i}.txt"), "fb=10500", "group=32"], stdout=subprocess.PIPE, env=child_env())
                             for i in range(ncores)]
                    inner = [float(pp.communicate()[0].split()[0]) for pp in procs]
                    wall = time.time() - t1
                line["cpu_baseline"] = {"value": round(sum(len(x) / t for t in inner) / 1e6, 3), "unit": "Msoftbits/s", "cores": ncores, "kind": "reference",
                                        "sample": f"{len(x)} soft bits (frames of channel 0, repeated) per core through the unmodified AeroL::processDemodulatedSoftBits "
                                                  f"in groups of 32, incl. its signal-unit parsing; one process per core ({wall:.1f} s wall)",
                                        "per_core": round(sum(len(x) / t for t in inner) / 1e6 / ncores, 3)}
            else:
                t1 = time.perf_counter()
                O.run_aerol(fb, x, 32)
                ct = time.perf_counter() - t1
                line["cpu_baseline"] = {"value": round(len(x) / ct / 1e6, 3), "unit": "Msoftbits/s", "cores": 1, "kind": "port",
                                        "sample": f"{len(x)} soft bits through oracle/aerol_oracle.c (AeroL::Decode restated), 32-bit groups, one thread"}
        emit(line)
    bank.close()
    finish(world)


def aerol_c_bench():
    """Row f4, Aero-L half: the 8400 bps C-channel bit pipeline (AeroL::DecodeC).  A step = one 4200-soft-bit frame (0.5 s) per channel,
    soft bits resident in HBM.  Algorithmic bytes per soft bit: 2 (int16 in) + 1.3 (depunctured block write, 5460 B per 4200) + 1.3
    (Viterbi read) + 0.65 + 0.65 (decoded bit write / read) + 1.3 (delay line r/w) = 7.2."""
    import torch
    import torch.distributed as dist

    from jaero_amd import aerol_frames as AF
    from jaero_amd import capi
    from jaero_amd import dist as jd
    from jaero_amd.demodulator import AeroLBank

    capi.lib()
    rank, world, local, dev, shared = setup_ranks()
    local = dev.index
    nch, K, W = ARGS.channels, ARGS.steps, ARGS.warmup
    flen, nuniq = 4200, 16
    streams = []
    for u in range(nuniq):
        frames, soft = AF.c_channel_case(9000 + u + 100 * rank, K + W + 1, 20.0, inv=(bool(u & 1), bool(u & 2)), lead=(u * 263) % flen)
        streams.append(soft[: (K + W) * flen])
    host = np.stack(streams)
    uniq = torch.from_numpy(host).to(dev)
    idx = torch.arange(nch, device=dev) % nuniq
    frames_t = torch.empty((K + W, nch, flen), dtype=torch.int16, device=dev)
    for i in range(K + W):
        frames_t[i].copy_(uniq[idx, i * flen:(i + 1) * flen])
    counts = torch.full((nch,), flen, dtype=torch.int32, device=dev)
    bank = AeroLBank(nch, 8400, device=local, max_softbits_per_write=flen, su_capacity=3 * (K + W) + 8)
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        bank.write_device(frames_t[i].data_ptr(), counts.data_ptr(), flen, flen, stream)

    dt, dts = run_timed(step, W, K, world, dev, before_timed=lambda: bank.profile_enable(True))
    names = ["bits", "viterbi", "post"]
    ms, nl = {}, {}
    for w, nm in enumerate(names):
        ms[nm], nl[nm] = bank.profile_read(w)
    oc, good = aerol_oracle_check("c", bank, nch, lambda c: host[c % nuniq], 3 * (K + W) + 8)
    nvoice = oc.get("voice_rows_compared", 0)
    if rank == 0:
        value = float(K) * flen * nch * world / dt / 1e6
        alg = 7.2
        # per kernel class and soft bit: bits = 2 B int16 in + 1 B into the frame buffer; viterbi = 5460 symbols in + 2714 bits out per 4200;
        # post = 2714 bits in, the delay line read and written, rows out
        alg_k = {"bits": 3.0, "viterbi": 1.95, "post": 2.25}
        dom = max(names, key=lambda k: ms[k])
        avg_ms = ms[dom] / K
        achieved_k = alg_k[dom] * flen * nch / (avg_ms * 1e-3) / 1e9
        kname = aerol_kernel_name(dom, nch, "c")
        tr, tr_from = aerol_traffic("pmc_summary_aerol_c.json", kname, nch)
        line = {
            "metric": "Msoftbits/s through the Aero-L C-channel bit pipeline (two-word unique word, 64x4 deinterleave, rate-3/4 depuncture, Viterbi, "
                      "delay line, descramble, sub-band signal units + voice bytes)",
            "value": round(value, 2), "unit": "Msoftbits/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{nch}-channel-per-GPU 8400 bps C-channel frames (4200 soft bits = 0.5 s per step and channel), {nuniq} distinct noisy "
                                   f"frame streams at different frame phases replicated over the channels, arm inversions mixed",
                       "channels_per_gpu": nch, "total_channels": nch * world, "frames_per_s": round(float(K) * nch * world / dt, 1),
                       "realtime_channel_equivalents": int(value * 1e6 / 8400),
                       "crc_clean_units_in_first_channels": good, "channels_checked": min(4, nch), "voice_frames_checked": nvoice,
                       "kernel_ms_per_step": {k: round(v / K, 4) for k, v in ms.items()}, "kernel_launches": nl},
            "roofline": {"bound": "hbm", "kernel": dom, "kernel_name": kname, "achieved": round(achieved_k, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved_k / HBM_PEAK_GBS, 6), "traffic": tr, "traffic_from": tr_from,
                         "alg_bytes_per_softbit": alg_k[dom], "softbits_per_launch": flen * nch, "avg_launch_ms": round(avg_ms, 4),
                         "whole_step": {"alg_bytes_per_softbit": alg, "achieved": round(alg * value * 1e6 / 1e9 / world, 2), "frac": round(alg * value * 1e6 / 1e9 / world / HBM_PEAK_GBS, 6)},
                         "note": "the dominant kernel of the step (its HIP-event time per working launch); integer work bound by VALU issue (the Viterbi: 225 wave "
                                 "instructions per trellis step for 64 blocks) and per-lane byte accesses, far below the HBM roof -- the figure is reported "
                                 "because the contract asks for it"},
        }
        line["config"].update(rank_fields(world, shared, dts, value * 1e6 * dt / world))
        aerol_check_or_exit(line, oc, "signal-unit / voice rows / events")
        if world == 1 and not ARGS.no_cpu_baseline:
            from oracle import oracle as O  # cpu_baseline leg only
            x = np.tile(host[0], max(1, int(1_000_000 / host.shape[1]) + 1))[:1_000_000]
            logical, physical, model = host_cores()
            if O.have_ref():
                with tempfile.TemporaryDirectory() as td:
                    path = os.path.join(td, "in.s16")
                    x.tofile(path)
                    t1 = time.time()
                    procs = [subprocess.Popen([O.REF_BIN, "aerol", path, os.path.join(td, f"o{i}.txt"), "fb=8400", "group=32"], stdout=subprocess.PIPE, env=child_env())
                             for i in range(logical)]
                    inner = [float(pp.communicate()[0].split()[0]) for pp in procs]
                    wall = time.time() - t1
                line["cpu_baseline"] = {"value": round(sum(len(x) / t for t in inner) / 1e6, 3), "unit": "Msoftbits/s", "cores": logical, "kind": "reference",
                                        "logical_cpus": logical, "physical_cores": physical,
                                        "sample": f"{len(x)} soft bits (frames of stream 0, repeated) per process through the unmodified AeroL::DecodeC in groups of 32; "
                                                  f"one process per logical CPU ({wall:.1f} s wall)"}
            else:
                t1 = time.perf_counter()
                a = O.AeroL(8400)
                for s_ in range(0, len(x), 32):
                    a.write(x[s_:s_ + 32])
                ct = time.perf_counter() - t1
                line["cpu_baseline"] = {"value": round(len(x) / ct / 1e6, 3), "unit": "Msoftbits/s", "cores": 1, "kind": "port",
                                        "sample": f"{len(x)} soft bits through oracle/aerol_oracle.c (DecodeC restated), 32-bit groups, one thread"}
        emit(line)
    bank.close()
    finish(world)


def msk_bench():
    """Row a2: the MSK demodulator (BASELINE configs[1] shape: synthetic 48 kHz 1200 bps MSK, scaled from 256 channels to a bank that
    fills the chip).  A step = one 4096-sample write for every channel (coarse 2^13 FFT every 2048 samples).  The MSK sample kernel keeps
    the newest 40 of its 80 matched-filter inputs in LDS (40 KiB per wavefront) and the older 40 in registers: four wavefronts per CU."""
    import torch
    import torch.distributed as dist

    from jaero_amd import capi, signalgen
    from jaero_amd import dist as jd
    from jaero_amd.demodulator import DemodulatorBank, MskSettings

    capi.lib()
    rank, world, local, dev, shared = setup_ranks()
    local = dev.index
    nch, chunk, K, W = ARGS.channels, ARGS.chunk, ARGS.steps, ARGS.warmup
    nsamp, nuniq = (K + W) * chunk, 32
    mfb = float(ARGS.fb)
    mbw = 1800.0 if mfb == 1200.0 else 900.0   # wide-bandwidth mode (JAERO/mainwindow.cpp:871-872) / the 600 bps default
    uniq = np.stack([signalgen.msk(nsamp, fb=mfb, fc=1000.0 + 7.0 * u, ebno_db=ARGS.ebno_db, seed=signalgen.SEED_BASE + 900 + u + 100 * rank)[0]
                     for u in range(nuniq)])  # [nuniq, nsamp]
    idx = torch.arange(nch, device=dev) % nuniq
    pcm = torch.from_numpy(np.ascontiguousarray(uniq.T)).to(dev)[:, idx].contiguous()  # frame-major [nsamp, nch], resident before the clock starts
    soft_cap = int(nsamp * mfb / 48000) + 64
    bank = DemodulatorBank(MskSettings(fb=mfb, lockingbw=mbw, freq_center=1000.0), nch, device=local, ebno=bool(ARGS.ebno), max_write_samples=chunk,
                           softbit_capacity=soft_cap)
    bank.set_flags(afc=False, sql=False, cpu_reduce=False)
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        bank.write(pcm[i * chunk:(i + 1) * chunk], layout=capi.PCM_FRAME_MAJOR, stream=stream)

    dt, dts = run_timed(step, W, K, world, dev, before_timed=lambda: bank.profile_enable(True))
    samp_ms, samp_n = bank.profile_read(0)
    coarse_ms, coarse_n = bank.profile_read(1)
    st = [bank.read_status(c) for c in range(min(8, nch))]
    if rank == 0:
        value = float(K) * chunk * nch * world / dt / 1e6
        dom = "sample_loop" if samp_ms >= coarse_ms else "coarse_freq"
        # algorithmic bytes per sample (MSK): PCM 2 + AGC ring r/w 16 + coarse ring write 16 + soft/state ~0.55 (+32 EbNo) = 34.55
        # (66.55); the two short delay lines (41 complex + 21 real entries per channel) are on-chip state in SURVEY's count -- this kernel
        # keeps them in HBM rows ([slot][lane], 48 B/sample of traffic): that is traffic ABOVE the algorithmic bytes, reported as such below.
        # coarse: (ring 128 KiB + y r/w 128 KiB) per 2048 samples = 128 B/sample
        per_sample = (34.55 + (32.0 if ARGS.ebno else 0.0)) if dom == "sample_loop" else 128.0
        dom_ms, launches = (samp_ms, samp_n) if dom == "sample_loop" else (coarse_ms, coarse_n)
        avg_ms = dom_ms / max(launches, 1)
        units = K * chunk * nch / max(launches, 1)
        achieved = per_sample * units / (avg_ms * 1e-3) / 1e9
        knames = {"sample_loop": bank.profile_kernel(0), "coarse_freq": bank.profile_kernel(1)}
        traffic, traffic_from = measured_traffic("pmc_summary_msk.json", dom, knames[dom], nch)
        line = {
            "metric": f"Msamples/s of real 48 kHz PCM through the {int(mfb)} bps MSK demodulator hot path", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{nch}-channel-per-GPU synthetic 48 kHz {int(mfb)} bps MSK continuous (BASELINE configs[1] shape, scaled from 256 channels), "
                                   f"{chunk}-sample writes, coarse 2^13 FFT every 2048 samples, AFC off, EbNo meters {'on' if ARGS.ebno else 'off'}, "
                                   f"Eb/N0 {ARGS.ebno_db} dB, {nuniq} distinct signals replicated over the channels",
                       "channels_per_gpu": nch, "total_channels": nch * world, "chunk": chunk, "realtime_channel_equivalents": int(value * 1e6 / 48000),
                       "locked_of_checked": int(sum(int(x.signal) for x in st)), "channels_checked": len(st),
                       "kernel_ms_total": {"sample_loop": round(samp_ms, 3), "coarse_freq": round(coarse_ms, 3)},
                       "kernel_ms_per_step": {"sample_loop": round(samp_ms / K, 4), "coarse_freq": round(coarse_ms / K, 4)},
                       "kernel_hbm_frac": {"sample_loop": round((34.55 + (32.0 if ARGS.ebno else 0.0)) * chunk * nch / (samp_ms / K * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                           "coarse_freq": round(128.0 * chunk * nch / (coarse_ms / K * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                       "delay_line_traffic_above_algorithmic_B_per_sample": 48.0,
                       "kernel_launches": {"sample_loop": samp_n, "coarse_freq": coarse_n}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_from": traffic_from, "kernel_name": knames[dom],
                         "alg_bytes_per_sample": per_sample, "samples_per_launch": units, "avg_launch_ms": round(avg_ms, 4)},
        }
        line["config"].update(rank_fields(world, shared, dts, float(K) * chunk * nch))
        line["config"]["kernels"] = knames
        if ARGS.check_channels > 0:
            from oracle import oracle as O  # checker only, after the clock has stopped
            try:
                O.build()
                oc = {"channels": [], "hard_bits_equal": True, "max_soft_byte_diff": 0, "bits_compared": 0}
                refs = {}
                for c in spread_channels(nch, ARGS.check_channels):
                    u = c % nuniq
                    if u not in refs:
                        refs[u] = O.run_demod(O.msk_settings(fb=mfb, lockingbw=mbw, freq_center=1000.0), uniq[u], chunk=chunk)
                    ref, got = refs[u], bank.read_softbits(c, cap=1 << 20)
                    n = len(ref["soft"])
                    ok = len(got) == n + ref["pending"] and bool(np.array_equal(got[:n] >= 128, ref["soft"] >= 128))
                    oc["hard_bits_equal"] &= ok
                    if ok and n:
                        oc["max_soft_byte_diff"] = max(oc["max_soft_byte_diff"], int(np.max(np.abs(got[:n].astype(int) - ref["soft"].astype(int)))))
                    oc["channels"].append(int(c))
                    oc["bits_compared"] += n
                line["config"]["oracle_check"] = oc
                if not oc["hard_bits_equal"]:
                    emit(line)
                    raise SystemExit("bench.py: hard decisions of a sampled MSK channel differ from the oracle's on the same PCM")
            except SystemExit:
                raise
            except Exception as e:
                line["config"]["oracle_check"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not ARGS.no_cpu_baseline:
            from oracle import oracle as O  # cpu_baseline leg only
            ncores = os.cpu_count() or 1
            n1 = min(ARGS.cpu_samples, 2_000_000)
            x, _ = signalgen.msk(n1, fb=mfb, fc=1000.0, ebno_db=ARGS.ebno_db, seed=signalgen.SEED_BASE + 77)
            if O.have_ref():
                with tempfile.TemporaryDirectory() as td:
                    path = os.path.join(td, "in.s16")
                    x.tofile(path)
                    t1 = time.time()
                    procs = [subprocess.Popen([O.REF_BIN, "time", "msk", path, f"chunk={chunk}", f"fb={int(mfb)}", f"lockingbw={int(mbw)}", "freq_center=1000"], stdout=subprocess.PIPE, env=child_env())
                             for _ in range(ncores)]
                    inner = [float(pp.communicate()[0].split()[0]) for pp in procs]
                    wall = time.time() - t1
                line["cpu_baseline"] = {"value": round(sum(n1 / t for t in inner) / 1e6, 3), "unit": "Msamples/s", "cores": ncores, "kind": "reference",
                                        "sample": f"{n1} samples of 48 kHz {int(mfb)} bps MSK per core through the unmodified MskDemodulator, {chunk}-sample writes, "
                                                  f"one process per core ({wall:.1f} s wall)"}
        emit(line)
    bank.close()
    finish(world)


def aerol_burst_bench():
    """Row f2: a step = what a burst demodulator bank emits for one burst per channel (start-of-burst marker, ~80 soft bits, unique word,
    a T packet of 7 signal units = 1472 channel bits, then noise up to 5248 entries = half a second), soft bits resident in HBM.  Metric: soft bits per
    second (and packets per second) through unique word search, trial deinterleave / Viterbi / CRC rounds."""
    import torch
    import torch.distributed as dist

    from jaero_amd import aerol_frames as AF
    from jaero_amd import capi
    from jaero_amd import dist as jd
    from jaero_amd.demodulator import AeroLBank

    capi.lib()
    rank, world, local, dev, shared = setup_ranks()
    local = dev.index
    nch, K, W = ARGS.channels, ARGS.steps, ARGS.warmup
    per, nuniq = 5248, 32  # the reference ignores a unique word for NumberOfBits-68 = 4924 soft bits after the previous one
    rng = np.random.default_rng(5 + rank)
    rb = lambda k: bytes(rng.integers(0, 256, k, dtype=np.uint8))
    streams = []
    for u in range(nuniq):
        segs = []
        for _ in range(K + W):
            x = AF.rt_burst_stream([("T", (rb(4), [rb(10) for _ in range(7)]))], gap=3600, lead=80, sigma=25.0, seed=int(rng.integers(1 << 30)),
                                   invert_i=bool(u & 1), invert_q=bool(u & 2))[64:]  # from the marker on
            segs.append(np.concatenate([x, np.full(per, 128, np.int16)])[:per])
        streams.append(np.concatenate(segs))
    host = np.stack(streams)
    soft = torch.from_numpy(host).to(dev)
    idx = torch.arange(nch, device=dev) % nuniq
    frames = torch.empty((K + W, nch, per), dtype=torch.int16, device=dev)
    for i in range(K + W):
        frames[i].copy_(soft[idx, i * per:(i + 1) * per])
    del soft
    counts = torch.full((nch,), per, dtype=torch.int32, device=dev)
    PROF = 4  # steps repeated after the clock and the oracle check, with the per-class events on (93 event pairs per step would weigh on the timed ones)
    bank = AeroLBank(nch, 10500, device=local, max_softbits_per_write=per, su_capacity=8 * (K + W + PROF) + 8, burst=True)
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        bank.write_device(frames[i].data_ptr(), counts.data_ptr(), per, per, stream)

    dt, dts = run_timed(step, W, K, world, dev)
    oc, npk = aerol_oracle_check("burst", bank, nch, lambda c: host[c % nuniq], 8 * (K + W) + 8)
    bank.profile_enable(True)
    for i in range(PROF):
        step(W + i % K)
    torch.cuda.synchronize()
    ms, nl = {}, {}
    for w, nm in enumerate(("bits", "viterbi", "post")):
        ms[nm], nl[nm] = bank.profile_read(w)
    vit = aerol_kernel_name("viterbi", nch, "b")
    if rank == 0:
        value = float(K) * per * nch * world / dt / 1e6
        # algorithmic bytes per soft bit: 2 (int16 in) + 1 (block write); every trial re-reads the block, deinterleaves and decodes it:
        # a T packet of n signal units is tried at 2, 5, 8 .. 5+3(n-1) columns = sum of 64*cols bytes read + written + read again
        trial_bytes = sum(3 * 64 * c for c in [2] + list(range(5, 5 + 3 * 6 + 1, 3)))
        alg = 3.0 + trial_bytes / float(per)
        line = {
            "metric": "Msoftbits/s through the Aero-L burst-mode (R/T channel) packet search", "value": round(value, 2), "unit": "Msoftbits/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{nch}-channel-per-GPU 10.5 kbps R/T bursts: per step and channel one burst as the burst demodulator emits it "
                                   f"(marker, lead-in, unique word, T packet with 7 signal units, noise; {per} entries), {nuniq} distinct streams",
                       "channels_per_gpu": nch, "total_channels": nch * world, "packets_per_s": round(float(K) * nch * world / dt, 1),
                       "packets_decoded_in_first_channels": npk, "channels_checked": min(4, nch), "expected": (K + W) * min(4, nch),
                       "kernel_ms_per_step": {k: round(v / PROF, 4) for k, v in ms.items()}, "kernel_launches_per_step": {k: nl[k] // PROF for k in nl},
                       "kernel_ms_from": f"{PROF} of the timed steps' inputs written again after the clock and the oracle check, event pairs around each class "
                                         "(bits = k_aerolb_bits + k_aerolb_deint)"},
            "roofline": {"bound": "hbm", "kernel": f"{vit} (trial decodes, one block per {'lane' if vit.endswith('lanes') else 'wavefront'})", "kernel_name": vit, "achieved": round(alg * value * 1e6 / 1e9, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg * value * 1e6 / 1e9 / HBM_PEAK_GBS, 6),
                         "traffic": aerol_traffic("pmc_summary_aerol_burst.json", vit, nch)[0], "traffic_from": aerol_traffic("pmc_summary_aerol_burst.json", vit, nch)[1],
                         "alg_bytes_per_softbit": round(alg, 2),
                         "note": "whole-step figure (the step is launch- and latency-bound: 31 rounds of 4 small kernels); the Viterbi trials "
                                 "are integer-VALU work, see the aerol workload"},
        }
        line["config"].update(rank_fields(world, shared, dts, value * 1e6 * dt / world))
        aerol_check_or_exit(line, oc, "R/T packets / events")
        if world == 1 and not ARGS.no_cpu_baseline:
            from oracle import oracle as O  # cpu_baseline leg only
            x = np.tile(host[0], max(1, int(1_000_000 / host.shape[1]) + 1))[:1_000_000]
            ncores = os.cpu_count() or 1
            if O.have_ref():
                with tempfile.TemporaryDirectory() as td:
                    path = os.path.join(td, "in.s16")
                    x.tofile(path)
                    t1 = time.time()
                    procs = [subprocess.Popen([O.REF_BIN, "aerol", path, os.path.join(td, f"o{i}.txt"), "fb=10500", "group=0", "burst=1"],
                                              stdout=subprocess.PIPE, env=child_env()) for i in range(ncores)]
                    inner = [float(pp.communicate()[0].split()[0]) for pp in procs]
                    wall = time.time() - t1
                line["cpu_baseline"] = {"value": round(sum(len(x) / t for t in inner) / 1e6, 3), "unit": "Msoftbits/s", "cores": ncores, "kind": "reference",
                                        "sample": f"{len(x)} soft bits (bursts of channel 0, repeated) per core through the unmodified AeroL in burst mode, "
                                                  f"demodulator-style groups; one process per core ({wall:.1f} s wall)"}
            else:
                t1 = time.perf_counter()
                O.run_aerol_burst(10500, x)
                ct = time.perf_counter() - t1
                line["cpu_baseline"] = {"value": round(len(x) / ct / 1e6, 3), "unit": "Msoftbits/s", "cores": 1, "kind": "port",
                                        "sample": f"{len(x)} soft bits through oracle/aerol_oracle.c in burst mode, one thread"}
        emit(line)
    bank.close()
    finish(world)


def host_cores():
    """(logical CPUs, physical cores, model string) of this box, from /proc/cpuinfo."""
    logical = os.cpu_count() or 1
    phys, model = set(), ""
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and not model:
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                pid = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":", 1)[1].strip()
            elif not ln.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return logical, (len(phys) or logical), model


def cpu_baseline_continuous(chunk: int, fb: float):
    """The unmodified reference OqpskDemodulator (oracle/_ref; the C port if that binary cannot run) on the host cores: first ONE
    process alone (what one core does), then one process per logical CPU (function-local statics make instances unshareable), each
    demodulating >= 10 s of the same kind of synthetic signal."""
    from jaero_amd import signalgen as G
    from oracle import oracle as O  # cpu_baseline leg only

    logical, physical, model = host_cores()
    n = max(ARGS.cpu_samples, 480000)
    pcm, _ = G.oqpsk(n, fb=fb, fc=8037.5, ebno_db=ARGS.ebno_db, seed=G.SEED_BASE + 77)
    use_ref = O.have_ref()
    if use_ref:
        try:
            subprocess.check_output([O.REF_BIN, "fft", "/dev/null", "/dev/null", "n=1"], stderr=subprocess.STDOUT, env=child_env())
        except Exception:
            use_ref = False
    kv = [f"chunk={chunk}"] + ([f"fb={int(fb)}", f"lockingbw={int(fb)}"] if fb != 10500 else [])
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "in.s16")
        pcm.tofile(path)

        def run(nproc):
            t0 = time.time()
            if use_ref:
                procs = [subprocess.Popen([O.REF_BIN, "time", "oqpsk", path] + kv, stdout=subprocess.PIPE, env=child_env()) for _ in range(nproc)]
            else:
                code = ("import sys,time,numpy as np; sys.path.insert(0,%r); from oracle import oracle as O; "
                        "x=np.fromfile(%r,dtype=np.int16); d=O.Demod(O.oqpsk_settings(fb=%r,lockingbw=%r)); t=time.time(); "
                        "[d.write(x[s:s+%d]) for s in range(0,len(x),%d)]; print(time.time()-t)") % (ROOT, path, fb, fb, chunk, chunk)
                procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, env=child_env()) for _ in range(nproc)]
            inner = [float(p.communicate()[0].split()[0]) for p in procs]
            return sum(n / t for t in inner) / 1e6, time.time() - t0

        single, w1 = run(1)
        # one process per physical core and one per logical CPU, three repeats each (the figure moves 40-57 Msamples/s between runs of the same
        # command on this pool: VERDICT r4 item 14); the better median is the quoted value, both sets are in the line
        sets = {}
        wall = 0.0
        for nproc in sorted({max(1, physical), max(1, logical)}):
            vals = []
            for _ in range(3):
                v, w = run(nproc)
                vals.append(v); wall += w
            vals.sort()
            sets[nproc] = vals
        best = max(sets, key=lambda k: sets[k][1])
        value = sets[best][1]
    return {"value": round(value, 3), "unit": "Msamples/s", "cores": best, "kind": "reference" if use_ref else "port",
            "sample": f"{n} samples ({n / 48000.0:.1f} s) of 48 kHz {fb / 1000:g}k OQPSK per process, {chunk}-sample writes, cpuReduce=false, "
                      f"{best} processes at once (median of three repeats; {wall:.1f} s wall for all {3 * len(sets)} runs)",
            "logical_cpus": logical, "physical_cores": physical, "cpu_model": model,
            "single_process_msps": round(single, 3), "per_process_msps_all_busy": round(value / best, 3),
            "repeats_msps": {str(k): [round(x, 3) for x in v] for k, v in sets.items()},
            "note": "FFT inside the reference build is the JFFT stand-in (oracle/ref/shim/jfft.h), not JFFT; with every core busy "
                    "the per-process rate drops (shared caches / SMT / 3.5 MB of rings per process); value = the better of one process per "
                    "physical core and one per logical CPU, repeats_msps holds the three runs of each (sorted)"}


def spread_channels(nch: int, k: int = 16):
    """Channel indices spread over a bank (wave edges, neighbouring waves, multiples of the CU count, the end)."""
    want = [0, 63, 64, 255, 256, 257, 511, 1023, 4095, 4096, 16383, 16384, 30000, 32767, 50001, 65535]
    got = sorted({c for c in want if c < nch} | {nch - 1})
    step = max(1, nch // k)
    for c in range(step // 2, nch, step):
        if len(got) >= k:
            break
        got = sorted(set(got) | {c})
    return got[:max(k, 1)] if len(got) > k else got


def oracle_check(O, fb, pcm_by_ch, n_pre, soft_by_ch, bits_by_ch, chunk):
    """Sampled channels of the bank against the oracle on the SAME PCM (from the first sample on): soft bits emitted after `n_pre`
    samples must have equal hard decisions (and bytes within 1); BER of both against the transmitted bits over that region."""
    res = {"channels": [], "hard_bits_equal": True, "max_soft_byte_diff": 0, "bits_compared": 0, "ber_gpu": 0.0, "ber_oracle": 0.0}
    errs_g = errs_o = nb = 0
    for c, x in pcm_by_ch.items():
        d = O.Demod(O.oqpsk_settings(fb=fb, lockingbw=fb))
        for s in range(0, n_pre, chunk):
            d.write(x[s:s + chunk])
        d.take_soft()
        p0 = d.pending  # soft bits of an incomplete group at the end of the pre-roll: the bank discarded them, the oracle emits them later
        for s in range(n_pre, len(x), chunk):
            d.write(x[s:s + chunk])
        ref = d.take_soft()[p0:]
        got = soft_by_ch[c]
        n = len(ref)
        ok = len(got) == n + d.pending and bool(np.array_equal(got[:n] >= 128, ref >= 128))
        res["hard_bits_equal"] &= ok
        if n and len(got) >= n:
            res["max_soft_byte_diff"] = max(res["max_soft_byte_diff"], int(np.max(np.abs(got[:n].astype(int) - ref.astype(int)))))
        res["channels"].append(int(c))
        res["bits_compared"] += n
        # BER: both output streams (imag first) against the transmitted arms, best lag / polarity per stream
        b = bits_by_ch[c]
        arms = (b[0::2], b[1::2])
        for name, soft in (("g", got[:n]), ("o", ref)):
            hard = (soft >= 128).astype(np.uint8)
            for stream in (hard[0::2], hard[1::2]):
                m = len(stream)
                if m < 1000:
                    continue
                best = m
                for arm in arms:
                    for lag in range(max(0, len(arm) - m - 1200), len(arm) - m + 1):
                        e = int(np.count_nonzero(arm[lag:lag + m] != stream))
                        best = min(best, e, m - e)
                if name == "g":
                    errs_g += best
                    nb += m
                else:
                    errs_o += best
    res["ber_gpu"] = errs_g / max(nb, 1)
    res["ber_oracle"] = errs_o / max(nb, 1)
    res["bits_scored"] = nb
    return res


def small_bank_run(kind: str, nch: int, chunk: int, K: int, W: int, dev, local, seed_offset: int = 0):
    """BASELINE configs as written (4096-channel OQPSK = configs[2], 256-channel 1200 bps MSK = configs[1]): Msamples/s of a short
    timed run (same clock discipline, inputs resident)."""
    import torch

    from jaero_amd import capi, signalgen
    from jaero_amd.demodulator import DemodulatorBank, MskSettings, OqpskSettings

    nsamp = (K + W) * chunk
    if kind == "oqpsk":
        st = signalgen.OqpskTorchStream(nch, nsamp, dev, ebno_db=ARGS.ebno_db, seed=signalgen.SEED_BASE + 31 + seed_offset, nphase=ARGS.timing_phases)
        pcm = st.render(0, nsamp)
        bank = DemodulatorBank(OqpskSettings(), nch, device=local, ebno=bool(ARGS.ebno), max_write_samples=chunk, softbit_capacity=int(nsamp * 10500 / 48000) + 64)
    elif kind == "burst_oqpsk":
        from jaero_amd.demodulator import BurstOqpskSettings

        pcm, _, _ = signalgen.burst_oqpsk_torch(nch, nsamp, dev, ebno_db=15.0, seed=signalgen.SEED_BASE + 61 + seed_offset)
        bank = DemodulatorBank(BurstOqpskSettings(), nch, device=local, max_write_samples=chunk, softbit_capacity=int(nsamp * 10500 / 48000) + 64)
    else:
        uniq = np.stack([signalgen.msk(nsamp, fb=1200.0, fc=1000.0 + 7.0 * u, ebno_db=ARGS.ebno_db, seed=signalgen.SEED_BASE + 900 + u)[0] for u in range(32)])
        idx = torch.arange(nch, device=dev) % 32
        pcm = torch.from_numpy(np.ascontiguousarray(uniq.T)).to(dev)[:, idx].contiguous()
        bank = DemodulatorBank(MskSettings(fb=1200.0, lockingbw=1800.0, freq_center=1000.0), nch, device=local, ebno=bool(ARGS.ebno), max_write_samples=chunk,
                               softbit_capacity=int(nsamp * 1200 / 48000) + 64)
    stream = torch.cuda.current_stream().cuda_stream
    for i in range(W):
        bank.write(pcm[i * chunk:(i + 1) * chunk], layout=capi.PCM_FRAME_MAJOR, stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(W, W + K):
        bank.write(pcm[i * chunk:(i + 1) * chunk], layout=capi.PCM_FRAME_MAJOR, stream=stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bank.close()
    v = float(K) * chunk * nch / dt / 1e6
    return {"channels": nch, "msamples_per_s": round(v, 2), "ms_per_step": round(dt / K * 1e3, 4), "realtime_channel_equivalents": int(v / 0.048)}


def burst_oracle_check(O, kind: str, pcm_by_ch: dict, soft_by_ch: dict, chunk: int):
    """Sampled channels of a burst bank against the oracle on the same PCM: the whole soft-bit stream (start-of-burst markers at the same
    places, equal hard decisions, bytes within 1)."""
    res = {"channels": [], "streams_equal": True, "max_soft_byte_diff": 0, "softbits_compared": 0, "bursts": 0}
    sett = O.burst_msk_settings(freq_center=1000.0, fb=1200.0) if kind == "burstmsk" else O.burst_oqpsk_settings()
    cache = {}
    for c, x in pcm_by_ch.items():
        key = x.tobytes()[:4096] + bytes(str(len(x)), "ascii")
        if key not in cache:
            cache[key] = O.run_burst(sett, x, chunk=chunk)["soft"]
        ref, got = cache[key], soft_by_ch[c]
        ok = len(got) == len(ref) and bool(np.array_equal(got == -1, ref == -1)) and bool(np.array_equal(got >= 128, ref >= 128))
        res["streams_equal"] &= ok
        if ok and len(ref):
            res["max_soft_byte_diff"] = max(res["max_soft_byte_diff"], int(np.max(np.abs(got.astype(int) - ref.astype(int)))))
        res["channels"].append(int(c))
        res["softbits_compared"] += int(len(ref))
        res["bursts"] += int((ref == -1).sum())
    return res


def burst_bench(msk: bool):
    """BASELINE configs[3] (burst OQPSK) / its MSK sibling: a step = one 4096-sample write of every channel of a burst bank."""
    import torch

    from jaero_amd import capi, signalgen
    from jaero_amd import dist as jd
    from jaero_amd.demodulator import BurstMskSettings, BurstOqpskSettings, DemodulatorBank

    capi.lib()
    rank, world, local, dev, shared = setup_ranks()
    nch, chunk, K, W = ARGS.channels, ARGS.chunk, ARGS.steps, ARGS.warmup
    lo, _ = jd.shard_range(nch * world, rank, world)
    stream = torch.cuda.current_stream().cuda_stream
    nsamp = (K + W) * chunk
    if msk:
        nuniq = 32
        rng = np.random.default_rng(signalgen.SEED_BASE + 900 + lo)
        uniq = np.stack([signalgen.burst_msk(nsamp, burst_starts=list(range(int(rng.integers(2000, 70000)), nsamp - 1000, 72000)), ndata=1000, fb=1200.0,
                                             fc=1000.0 + float(rng.uniform(-8, 8)), ebno_db=18.0, seed=signalgen.SEED_BASE + 900 + lo + u)[0] for u in range(nuniq)])
        ut = torch.from_numpy(uniq).to(dev)                                  # [nuniq][nsamp]
        idx = torch.arange(nch, device=dev) % nuniq
        pcm = ut.t().contiguous()[:, idx].contiguous()                       # frame-major [nsamp][nch]
        bank = DemodulatorBank(BurstMskSettings(freq_center=1000.0, fb=1200.0), nch, device=dev.index, max_write_samples=chunk,
                               softbit_capacity=int(nsamp * 1200 / 48000) + 64)
    else:
        pcm, _, _ = signalgen.burst_oqpsk_torch(nch, nsamp, dev, ebno_db=15.0, seed=signalgen.SEED_BASE + lo)
        bank = DemodulatorBank(BurstOqpskSettings(), nch, device=dev.index, max_write_samples=chunk, softbit_capacity=int(nsamp * 10500 / 48000) + 64)
    bank.set_flags(afc=False, sql=False, cpu_reduce=False)

    def step(i):
        bank.write(pcm[i * chunk:(i + 1) * chunk], layout=capi.PCM_FRAME_MAJOR, stream=stream)

    dt, dts = run_timed(step, W, K, world, dev, before_timed=lambda: bank.profile_enable(True))
    extra = rank_fields(world, shared, dts, float(K) * chunk * nch)
    if rank == 0 and ARGS.check_channels > 0:
        from oracle import oracle as O  # checker only, after the clock has stopped
        try:
            O.build()
            check = spread_channels(nch, ARGS.check_channels)
            pcm_by_ch = {c: pcm[:, c].cpu().numpy() for c in check}
            soft_by_ch = {c: bank.read_softbits(c, cap=1 << 20) for c in check}
            oc = burst_oracle_check(O, "burstmsk" if msk else "burstoqpsk", pcm_by_ch, soft_by_ch, chunk)
            extra["oracle_check"] = oc
            if not oc["streams_equal"]:
                extra["oracle_check"]["FAILED"] = True
        except Exception as e:
            extra["oracle_check"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    burst_line(bank, rank, world, nch, chunk, K, W, dt, float(K) * chunk * nch * world / dt / 1e6, msk=msk, extra=extra)
    failed = bool(extra.get("oracle_check", {}).get("FAILED"))
    bank.close()
    finish(world)
    if failed:
        raise SystemExit("bench.py: a sampled channel's soft-bit stream differs from the oracle's on the same PCM")


def edge_c_abi_check(rank, world, dev, timeout_s=60.0):
    """First-contact insurance for a node with one GPU per rank: the two edge operations through the C ABI itself
    (jaero_comm_create / jaero_fan_out_pcm / jaero_gather_softbits: RCCL grouped send / recv loaded by dlopen inside libjaero_hip, not
    torch.distributed) on a small bank of 64 channels per rank, checked value by value.  Runs in a side thread with a hard timeout AFTER the
    timed region: a hung RCCL group then costs this entry (ok = false, "timeout"), never the measured line."""
    import ctypes as C
    import threading

    import torch
    import torch.distributed as dist

    from jaero_amd import capi
    from jaero_amd import dist as jd

    res = {"ok": False}

    def work():
        L = capi.lib()
        t0 = time.perf_counter()
        ident = (C.c_char * 128)()
        if rank == 0:
            capi.check(L.jaero_comm_get_unique_id(ident))
        t = torch.tensor(list(ident.raw), dtype=torch.uint8, device=dev)
        dist.broadcast(t, 0)
        ident = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().tolist()))
        comm = C.c_void_p()
        capi.check(L.jaero_comm_create(dev.index, rank, world, ident, C.byref(comm)))
        try:
            per, nsamp, cap = 64, 1024, 96
            total = per * world
            lo, hi = jd.shard_range(total, rank, world)
            st = torch.cuda.current_stream().cuda_stream
            srow = torch.arange(nsamp, device=dev, dtype=torch.int32)[:, None]
            ccol = torch.arange(total, device=dev, dtype=torch.int32)[None, :]
            frames_all = ((srow * 31 + ccol * 7) & 0x7FFF).to(torch.int16).contiguous()   # every rank can form the whole pattern: the check is local
            mine = torch.zeros((nsamp, hi - lo), dtype=torch.int16, device=dev)
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            capi.check(L.jaero_fan_out_pcm(comm, 0, frames_all.data_ptr() if rank == 0 else None, nsamp, total, mine.data_ptr(), st))
            e1.record()
            soft = ((torch.arange(hi - lo, device=dev, dtype=torch.int32)[:, None] + lo) * 3 + torch.arange(cap, device=dev, dtype=torch.int32)[None, :]).to(torch.int16).contiguous()
            cnt = (torch.arange(hi - lo, device=dev, dtype=torch.int32) + lo) % cap
            soft_all = torch.zeros((total, cap), dtype=torch.int16, device=dev)
            cnt_all = torch.zeros((total,), dtype=torch.int32, device=dev)
            capi.check(L.jaero_gather_softbits(comm, 0, soft.data_ptr(), cnt.data_ptr(), total, cap, soft_all.data_ptr() if rank == 0 else None,
                                               cnt_all.data_ptr() if rank == 0 else None, st))
            e2.record()
            torch.cuda.synchronize()
            ok = bool(torch.equal(mine, frames_all[:, lo:hi]))
            if rank == 0:
                want_soft = (torch.arange(total, device=dev, dtype=torch.int32)[:, None] * 3 + torch.arange(cap, device=dev, dtype=torch.int32)[None, :]).to(torch.int16)
                ok = ok and bool(torch.equal(soft_all, want_soft)) and bool(torch.equal(cnt_all, (torch.arange(total, device=dev, dtype=torch.int32) % cap)))
            res.update({"ok_this_rank": ok, "fan_out_ms": round(e0.elapsed_time(e1), 3), "gather_ms": round(e1.elapsed_time(e2), 3),
                        "ms": round((time.perf_counter() - t0) * 1e3, 1), "channels": total, "samples": nsamp})
        finally:
            L.jaero_comm_destroy(comm)
        f = torch.tensor([1 if res.get("ok_this_rank") else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        res["ok"] = bool(int(f.item()))

    def guarded():
        try:
            work()
        except Exception as e:
            res["error"] = f"{type(e).__name__}: {e}"[:300]

    th = threading.Thread(target=guarded, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        res.update({"ok": False, "error": f"timeout: the C-ABI edge operations did not return within {timeout_s:.0f} s"})
        res["hung"] = True
    return res


# the other workloads of SURVEY 8 rows (a) / (f), run behind the headline by the ONE driver command
OTHER_WORKLOADS = [("msk_1200", ["--workload", "msk"], None), ("msk_600", ["--workload", "msk", "--fb", "600"], None),
                   ("oqpsk8400", ["--workload", "oqpsk8400", "--preroll", "40"], None),
                   # a burst OQPSK channel carries one burst per second = 11.7 writes, a burst MSK channel one per 1.5 s = 17.6 writes: the timed steps
                   # cover one whole burst period (with fewer, the figure depends on where in its period a channel is)
                   ("burst_oqpsk", ["--workload", "burst_oqpsk"], 12), ("burst_msk", ["--workload", "burst_msk"], 18),
                   # (the P channels of the synthetic bank acquire lock during their first three frames: four warm-up frames, or the first timed step is an
                   # acquisition step twice as long as the others -- round 6: 4.26 ms mean over six steps against a 3.35 ms median)
                   ("aerol", ["--workload", "aerol", "--warmup", "4"], None), ("aerol_burst", ["--workload", "aerol_burst"], None), ("aerol_c", ["--workload", "aerol_c"], None)]


def summarise_workload(d: dict) -> dict:
    """the few figures of another workload's own line that go into the headline's line"""
    r, c = d.get("roofline") or {}, d.get("config") or {}
    oc = c.get("oracle_check")
    if isinstance(oc, dict):
        flag = [v for k, v in oc.items() if k in ("hard_bits_equal", "streams_equal", "rows_equal", "events_equal", "packets_equal", "voice_equal") and isinstance(v, bool)]
        ocs = {"ok": bool(flag) and all(flag), "channels": len(oc.get("channels", []))}
        for k in ("max_soft_byte_diff", "bits_compared", "softbits_compared", "rows_compared", "bursts", "error"):
            if k in oc:
                ocs[k] = oc[k]
    else:
        ocs = {"ok": False, "error": "no oracle check in the workload's line"}
    g, cal = d.get("gpu_state") or {}, d.get("calib") or {}
    out = {"value": d.get("value"), "unit": d.get("unit"), "ms_per_step": d.get("ms_per_step"), "steps": d.get("steps"),
           "dominant_kernel": r.get("kernel_name") or r.get("kernel"), "kernel_ms": r.get("avg_launch_ms") or r.get("kernel_ms_per_step") or r.get("ms"),
           "bound": r.get("bound"), "frac": r.get("frac"), "oracle_check": ocs,
           "sclk_mhz": g.get("sclk_mhz_mean"), "power_w": g.get("power_w_mean"), "throttled": g.get("throttled"), "calib_fp64_tflops": cal.get("fp64_tflops")}
    if r.get("frac_of_calib_hbm") is not None:
        out["frac_of_calib_hbm"] = r.get("frac_of_calib_hbm")
    if r.get("floor_ms") is not None:
        out["issue_floor_ms"] = r.get("floor_ms")
    return out


def other_workloads_pass(channels: int):
    """Each of the other workloads as its own process on the same device (its own bank, its own resident input, its own oracle check after its
    clock has stopped), `--other-steps` timed steps; returns {name: summary}.  The long lines go to gpurun_out/bench_details.json."""
    out = {}
    t_all = time.time()
    for name, args, steps in OTHER_WORKLOADS:
        cmd = [sys.executable, os.path.abspath(__file__)] + ["--warmup", "2"] + args + ["--steps", str(steps or ARGS.other_steps), "--channels", str(channels), "--no-cpu-baseline",
                                                                    "--as-written", "0", "--no-other-workloads", "--sustain", "0", "--gpus", "1"]
        t0 = time.time()
        try:
            env = dict(os.environ)
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
            lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
            if not lines:
                out[name] = {"error": ("rc %d: " % r.returncode) + (r.stderr.strip().splitlines() or ["no output"])[-1][:300]}
            else:
                out[name] = summarise_workload(json.loads(lines[-1]))
                if r.returncode != 0:
                    out[name]["rc"] = r.returncode
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out[name]["wall_s"] = round(time.time() - t0, 1)
    out["wall_s_total"] = round(time.time() - t_all, 1)
    return out


def main():
    """Continuous OQPSK: 10.5 kbps (the headline, BASELINE configs[2] shape; at N > 1 configs[4]) or 8400 bps (row f4)."""
    import torch
    import torch.distributed as dist

    from jaero_amd import capi, signalgen
    from jaero_amd import dist as jd
    from jaero_amd.demodulator import DemodulatorBank, OqpskSettings

    capi.lib()  # fail loudly if the HIP extension is missing
    rank, world, local, dev, shared = setup_ranks()
    local = dev.index
    nch, chunk, K, W = ARGS.channels, ARGS.chunk, ARGS.steps, ARGS.warmup
    lo, _ = jd.shard_range(nch * world, rank, world)
    stream = torch.cuda.current_stream().cuda_stream

    fb = 8400.0 if ARGS.workload == "oqpsk8400" else 10500.0
    # Untimed pre-roll (part of preparing the resident state, like the W warm-up steps): the reference's AGC averages over 4 s and
    # its BER settles once that window has filled ("score BER after t = 4 s"), so the bank is run to t >= 4 s before
    # the clock starts, and the bits of the timed steps are the ones scored.
    pre = ARGS.preroll if ARGS.preroll >= 0 else max(0, int(np.ceil(4.0 * 48000 / chunk)) - W)
    # Sustain: the LAST S steps of the pre-roll are resident beside the warm-up and timed steps and run back to back with
    # them, so that the device has been at this load for ~--sustain seconds when the clock starts (a step of a 65536-channel bank is ~23 ms).
    est_step_s = 0.023 * max(nch, 4096) / 65536.0 * (1.5 if fb == 8400 else 1.0)
    S = int(np.ceil(ARGS.sustain / est_step_s)) if ARGS.sustain > 0 else 0
    check = spread_channels(nch, ARGS.check_channels) if ARGS.check_channels > 0 else []
    cidx = torch.tensor(check, dtype=torch.long, device=dev)
    host_pcm = {c: [] for c in check}

    def keep(block):  # the sampled channels' PCM goes to the host for the oracle leg (after the clock has stopped)
        if check:
            cols = block[:, cidx].t().contiguous().cpu().numpy()
            for j, c in enumerate(check):
                host_pcm[c].append(cols[j])

    free0, hbm_total = torch.cuda.mem_get_info(dev.index)
    bank = DemodulatorBank(OqpskSettings(fb=fb, lockingbw=fb, coarsefreqest_fft_power=14), nch, device=local,
                           ebno=bool(ARGS.ebno), max_write_samples=chunk, softbit_capacity=int((W + K) * chunk * fb / 48000) + 64)
    free1, _ = torch.cuda.mem_get_info(dev.index)
    bytes_per_channel = max(free0 - free1, 1) / float(nch)
    step_bytes = chunk * nch * 2
    # resident steps: what fits beside the bank with room for the generator's temporaries, the calibration buffers (8 GB) and the small banks
    S = max(0, min(S, int((free1 - (24 << 30)) * 0.6 / step_bytes) - (W + K)))
    pre = max(pre, S)
    total = (pre + W + K) * chunk
    gen = signalgen.OqpskTorchStream(nch, total, dev, fb=fb, ebno_db=ARGS.ebno_db, seed=signalgen.SEED_BASE + lo, nphase=ARGS.timing_phases)
    bank.set_flags(afc=False, sql=False, cpu_reduce=False)
    PB = 8  # rendered (and, for the non-resident part of the pre-roll, consumed) 8 steps at a time
    for b in range(0, pre - S, PB):
        nb = min(PB, pre - S - b)
        blk = gen.render(b * chunk, nb * chunk)
        keep(blk)
        for i in range(nb):
            bank.write(blk[i * chunk:(i + 1) * chunk], layout=capi.PCM_FRAME_MAJOR, stream=stream)
        torch.cuda.synchronize()
        bank.discard_softbits(stream)
        del blk
    # sustain + warm-up + timed steps: resident in HBM before the clock starts
    nres = S + W + K
    pcm = torch.empty((nres * chunk, nch), dtype=torch.int16, device=dev)
    for b in range(0, nres, PB):
        nb = min(PB, nres - b)
        blk = gen.render((pre - S + b) * chunk, nb * chunk)
        pcm[b * chunk:(b + nb) * chunk] = blk
        keep(blk)
        del blk
    torch.cuda.synchronize()

    def step(i):
        bank.write(pcm[i * chunk:(i + 1) * chunk], layout=capi.PCM_FRAME_MAJOR, stream=stream)
        if i < S:
            bank.discard_softbits(stream)  # the soft bits of the sustain steps (the warm-up's and the timed steps' are the ones kept and checked)

    dt, dts = run_timed(step, W, K, world, dev, before_timed=lambda: bank.profile_enable(True), sustain=S)

    samp_ms, samp_n = bank.profile_read(0)
    coarse_ms, coarse_n = bank.profile_read(1)
    kernel_names = {"sample_loop": bank.profile_kernel(0), "coarse_freq": bank.profile_kernel(1)}
    total_samples = float(K) * chunk * nch * world
    value = total_samples / dt / 1e6
    soft_by_ch = {c: bank.read_softbits(c, cap=1 << 22) for c in check}
    locked = sum(int(bank.read_status(c).signal) for c in check)

    edge = None
    if world > 1 and shared:
        edge = {"skipped": "ranks share a device: the control plane is gloo, the RCCL edge operations need one GPU per rank"}
    elif world > 1:
        # the two edge operations of the path (north_star: "RCCL over xGMI used only to fan out shared IQ and gather decoded bits"),
        # once, untimed for the metric: rank 0 scatters one step of frames for ALL ranks' channels, rank 0 gathers every channel's
        # soft bits of the run
        try:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            frames = None
            if rank == 0:
                frames = pcm[:chunk].repeat(1, world) if world * nch * chunk * 2 < (8 << 30) else pcm[:chunk // 8].repeat(1, world)
            ns = chunk if world * nch * chunk * 2 < (8 << 30) else chunk // 8
            torch.cuda.synchronize(); dist.barrier()
            ev[0].record()
            mine = jd.fan_out_pcm(frames, nch * world, ns, src=0, device=dev)
            ev[1].record()
            cap = 256
            soft_t = torch.zeros((nch, cap), dtype=torch.int16, device=dev)
            cnt_t = torch.full((nch,), cap, dtype=torch.int32, device=dev)
            ev[2].record()
            sa, ca = jd.gather_softbits(soft_t, cnt_t, nch * world, dst=0)
            ev[3].record()
            torch.cuda.synchronize()
            edge = {"fan_out_pcm_ms": round(ev[0].elapsed_time(ev[1]), 3), "fan_out_bytes": int(ns * nch * (world - 1) * 2),
                    "gather_softbits_ms": round(ev[2].elapsed_time(ev[3]), 3), "gather_bytes": int(nch * (world - 1) * (cap * 2 + 4)),
                    "backend": "nccl (RCCL), point-to-point send/recv", "shape_ok": bool(mine.shape == (ns, nch))}
        except Exception as e:  # an edge operation that fails must not take the measured line with it
            edge = {"error": f"{type(e).__name__}: {e}"[:300]}
        dist.barrier()
    bank.close()
    del pcm
    torch.cuda.empty_cache()

    # BASELINE configs as written: configs[2] / configs[1] at N = 1, configs[4] (32768 channels over 8 GPUs = 4096 per GPU) at N > 1
    as_written = None
    if ARGS.as_written and fb == 10500:
        try:
            if world == 1:
                as_written = {"configs[2] 4096-channel 10.5 kbps OQPSK": small_bank_run("oqpsk", 4096, chunk, K, W, dev, local),
                              "configs[1] 256-channel 1200 bps MSK": small_bank_run("msk", 256, chunk, K, W, dev, local),
                              "configs[3] 4096-channel 10.5 kbps burst OQPSK": small_bank_run("burst_oqpsk", 4096, chunk, K, W, dev, local),
                              "note": "banks as BASELINE.json words them: 64 (4) wavefronts on 1024 SIMDs, latency bound (DESIGN 17, profiles/r5_small_bank.md)"}
            else:
                r = small_bank_run("oqpsk", 4096, chunk, K, W, dev, local, seed_offset=lo)
                t = torch.tensor([r["ms_per_step"]], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                v = 4096.0 * world * chunk / (float(t.item()) * 1e-3) / 1e6
                as_written = {f"configs[4] {4096 * world}-channel 10.5 kbps OQPSK sharded over {world} GPUs (4096 per GPU)":
                              {"channels": 4096 * world, "msamples_per_s": round(v, 2), "ms_per_step_slowest_rank": round(float(t.item()), 4),
                               "realtime_channel_equivalents": int(v / 0.048)},
                              "note": "BASELINE configs[4] as written (32768 channels over 8 GPUs): 64 wavefronts per GPU on 1024 SIMDs, the small-bank rate"}
        except Exception as e:
            as_written = {"error": f"{type(e).__name__}: {e}"[:300]}

    # The C-ABI edge operations LAST among the things that need every rank (the as-written banks above still use the control plane): if RCCL hangs in there,
    # every rank times out in the check's closing all-reduce, rank 0 prints the measured line with the failure in it, and all ranks leave without a barrier.
    edge_c = None
    if world > 1 and shared:
        edge_c = {"ok": None, "skipped": "ranks share a device (RCCL refuses two ranks on one GPU)"}
    elif world > 1:
        edge_c = edge_c_abi_check(rank, world, dev)

    if rank == 0:
        with_eb = bool(ARGS.ebno)
        b_samp = ALG_BYTES_SAMPLE_KERNEL + (ALG_BYTES_SAMPLE_KERNEL_EBNO if with_eb else 0.0)
        if fb == 8400:
            b_samp += 16.0  # the prefiltered complex sample the loop reads instead of forming PCM x mixer2 (k_pre8400_fir writes it: charged there)
        dom = "sample_loop" if samp_ms >= coarse_ms else "coarse_freq"
        per_sample = b_samp if dom == "sample_loop" else ALG_BYTES_COARSE_KERNEL
        dom_ms = samp_ms if dom == "sample_loop" else coarse_ms
        ms_per_step = dom_ms / K                       # this kernel's time per step (HIP events on the launch stream inside jaero_write)
        samples_per_step = float(chunk) * nch           # what its launches of one step process together (this rank)
        achieved = per_sample * samples_per_step / (ms_per_step * 1e-3) / 1e9
        traffic, traffic_from = measured_traffic("pmc_summary.json" if fb != 8400 else "pmc_summary_8400.json", dom, kernel_names[dom], nch)
        # fp64 side roof: the reference's arithmetic is ~350 fp64 flops per sample in the sample loop and 3 x 5 N log2 N
        # per estimate (N = 2^14, one estimate per 4096 samples and channel) = 840 per sample; 78.6 TFLOP/s fp64 vector peak
        fl_samp, fl_coarse = 350.0, 3.0 * 5.0 * 16384 * 14 / 4096.0
        fp64 = {"bound": "fp64_valu", "peak": 78.6e12, "unit": "FLOP/s",
                "achieved": round((fl_samp + fl_coarse) * value * 1e6 / world, 1), "frac": round((fl_samp + fl_coarse) * value * 1e6 / world / 78.6e12, 5),
                "flops_per_sample": {"sample_loop": fl_samp, "coarse_freq": fl_coarse},
                "per_kernel_frac": {"sample_loop": round(fl_samp * samples_per_step / (samp_ms / K * 1e-3) / 78.6e12, 5),
                                    "coarse_freq": round(fl_coarse * samples_per_step / (coarse_ms / K * 1e-3) / 78.6e12, 5) if coarse_ms else None},
                "note": "algorithmic flops (FMA = 2), not issued instructions: see roofline_fp64_issue"}
        issue = issue_roofline({"sample_loop": samp_ms / K, "coarse_freq": coarse_ms / K}, kernel_names, nch)
        name = "10.5 kbps" if fb == 10500 else "8400 bps (C channel)"
        cfg_name = ("BASELINE configs[2] shape" if world == 1 else f"BASELINE configs[4] shape: {nch * world} channels sharded over {world} GPUs")
        line = {
            "metric": f"Msamples/s of real 48 kHz PCM through the {name} OQPSK demodulator hot path",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{nch}-channel-per-GPU synthetic 48 kHz {name} OQPSK continuous ({cfg_name}, scaled to {nch} channels per GPU = one "
                                   f"wavefront of 64 channels per SIMD), {chunk}-sample writes, coarse 2^14 FFT every 4096 samples, AFC off, EbNo meters "
                                   f"{'on' if with_eb else 'off'}, Eb/N0 {ARGS.ebno_db} dB, per-channel carrier (8000 +- 100 Hz), bits, noise, symbol phase; "
                                   f"{pre} untimed pre-roll steps (the last {S} resident and back to back with the warm-up: sustain), timed steps start at "
                                   f"t = {(pre + W) * chunk / 48000.0:.2f} s of signal",
                       "channels_per_gpu": nch, "total_channels": nch * world, "chunk": chunk, "ebno_meters": with_eb,
                       "preroll_steps": pre, "timing_phases": ARGS.timing_phases,
                       "realtime_channel_equivalents": int(value / 0.048),
                       "realtime_channel_equivalents_note": "throughput / 48 kS/s, NOT channels resident at once (see resident_channels_max)",
                       "hbm_bytes_per_resident_channel": int(bytes_per_channel),
                       "resident_channels_max": int(hbm_total / bytes_per_channel),
                       "whole_path_hbm_frac_at_163B_per_sample": round(value * 1e6 * ALG_BYTES_WHOLE_PATH / 1e9 / (HBM_PEAK_GBS * world), 5),
                       "kernel_ms_per_step": {"sample_loop": round(samp_ms / K, 4), "coarse_freq": round(coarse_ms / K, 4)},
                       "kernel_ms_total": {"sample_loop": round(samp_ms, 3), "coarse_freq": round(coarse_ms, 3)},
                       "kernel_launches": {"sample_loop": samp_n, "coarse_freq": coarse_n}, "kernels": kernel_names,
                       "locked_of_checked": locked, "channels_checked": len(check)},
            "roofline": {"bound": "hbm", "kernel": dom, "kernel_name": kernel_names[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_from": traffic_from,
                         "alg_bytes_per_sample": per_sample, "samples_per_step": samples_per_step, "kernel_ms_per_step": round(ms_per_step, 4),
                         "launches_per_step": round((samp_n if dom == "sample_loop" else coarse_n) / float(K), 2),
                         "note": "achieved = alg_bytes_per_sample x samples_per_step / kernel_ms_per_step (HIP events on the launch stream)"},
            "roofline_fp64": fp64,
            "roofline_fp64_issue": issue,
        }
        line["config"].update(rank_fields(world, shared, dts, float(K) * chunk * nch))
        if edge:
            line["config"]["edge_collectives"] = edge
        if edge_c:
            line["config"]["edge_collectives_c_abi"] = edge_c
        if as_written:
            line["config"]["as_written"] = as_written
        if world == 1 and fb == 10500 and not ARGS.no_other_workloads:
            try:
                line["config"]["other_workloads"] = other_workloads_pass(nch)
            except Exception as e:
                line["config"]["other_workloads"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if check:
            from oracle import oracle as O  # checker only, after the clock has stopped
            try:
                O.build()
                pcm_by_ch = {c: np.concatenate(host_pcm[c]) for c in check}
                bits_by_ch = {c: gen.bits_of(c) for c in check}
                oc = oracle_check(O, fb, pcm_by_ch, pre * chunk, soft_by_ch, bits_by_ch, chunk)
                line["config"]["oracle_check"] = oc
                assert oc["hard_bits_equal"], "hard decisions of a sampled channel differ from the oracle's on the same PCM"
            except AssertionError:
                emit(line)
                raise
            except Exception as e:
                line["config"]["oracle_check"] = {"error": str(e)}
        if world == 1 and not ARGS.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline_continuous(chunk, fb)
            except Exception as e:  # never lose the GPU line because the CPU leg failed
                line["cpu_baseline"] = {"value": None, "error": str(e)}
        emit(line)
    if edge_c and edge_c.get("hung"):
        os._exit(0)  # a side thread is stuck inside RCCL: the line is out (rank 0), do not wait for a barrier that cannot complete
    finish(world)


if __name__ == "__main__":
    ARGS = parse()
    maybe_spawn()
    if ARGS.workload in ("burst_oqpsk", "burst_msk"):
        burst_bench(ARGS.workload == "burst_msk")
    elif ARGS.workload == "msk":
        pass  # 65536 channels: four wavefronts per CU (40 KiB of LDS each, the older half of the filter history in registers)
        msk_bench()
    elif ARGS.workload == "aerol":
        aerol_bench()
    elif ARGS.workload == "aerol_c":
        aerol_c_bench()  # 65536 channels as the other workloads (16384: a quarter of the SIMDs busy in the lane-per-channel kernels, 9.3 Gsoftbits/s)
    elif ARGS.workload == "aerol_burst":
        aerol_burst_bench()  # 65536 channels (16384: 11.9 Gsoftbits/s, the step is launch- and latency-bound)
    else:
        main()
