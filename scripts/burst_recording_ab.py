#!/usr/bin/env python3
"""The WHOLE of the reference's 10.5 kbps R/T-channel recording (samples/10.5k_burst_sample.mp3, 33.7 s, 12 bursts) through reference, oracle and GPU,
and the A/B the verdict of round 5 asked for: does the continuous kernels' arithmetic (matched filter op for op, glibc's hypot, correctly rounded
atan2) move anything in the burst OQPSK tracking kernel ON AN OFF-AIR SIGNAL?  (profiles/r5_burst_ab.json answered it on synthetic bursts only.)

    python scripts/burst_recording_ab.py stage      build container: decode (scripts/mp3_decode.py), resample to 48 kHz -> gpurun_stage/ (not committed)
    python scripts/burst_recording_ab.py cpu        build container: unmodified reference against the oracle on the whole recording
    python scripts/burst_recording_ab.py gpu        GPU box: a bank fed the recording from four starting points against the oracle; run once with the product
                                                    and once with JAERO_HIP_LIB=gpurun_tmp/libjaero_hip_burstexact.so (make -C jaero_amd/csrc ab_burst)
Each prints one JSON line (kept under profiles/)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAGE = os.path.join(ROOT, "gpurun_stage", "recording_burst_oqpsk_10k5_48k.i16")


def stage():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_burst_recording_golden as mk

    pcm, info = mk.decode_48k(None)
    os.makedirs(os.path.dirname(STAGE), exist_ok=True)
    pcm.tofile(STAGE)
    print(json.dumps({"staged": STAGE, "samples": int(len(pcm)), "seconds": len(pcm) / 48000.0, "decoder": info}))


def cpu():
    from oracle import oracle as O

    pcm = np.fromfile(STAGE, dtype=np.int16)
    r = O.run_ref("burstoqpsk", pcm)
    o = O.run_burst(O.burst_oqpsk_settings(), pcm, chunk=4096)
    pk, bad, _ = O.run_ref_aerol_burst(10500, r["soft"])
    a = O.run_aerol_burst(10500, r["soft"])
    got = O.packets_from_rows(a["packets"])
    same = len(got) == len(pk) and all((t == 1 and p[0] == "R" and d[:17] == p[1]) or
                                       (t == 2 and p[0] == "T" and d[:4] == p[1] and [d[6 + 12 * j: 16 + 12 * j] for j in range(len(p[3]))] == p[3])
                                       for (t, d), p in zip(got, pk))
    print(json.dumps({"what": "unmodified reference against the restatement on the whole burst recording", "samples": int(len(pcm)), "seconds": len(pcm) / 48000.0,
                      "soft_bits": int(len(r["soft"])), "bursts": int((r["soft"] == -1).sum()), "soft_bits_identical": bool(np.array_equal(r["soft"], o["soft"])),
                      "packets_printed_by_the_reference": [[p[0], len(p[3]) if p[0] == "T" else 0] for p in pk], "bad_packet_lines": bad,
                      "oracle_bad_packet_events": int((a["events"][:, 1] == 3).sum()), "packets_identical": bool(same)}))


def gpu():
    from jaero_amd import capi
    from jaero_amd import demodulator as B
    from oracle import oracle as O

    capi.lib()
    full = np.fromfile(STAGE, dtype=np.int16)
    shifts = [0, 1234, 7777, 20001]
    n = (len(full) - max(shifts)) // 4096 * 4096
    pcm = np.stack([full[s:s + n] for s in shifts])
    nch = len(shifts)
    bank = B.DemodulatorBank(B.BurstOqpskSettings(), nch, device=0, capture_symbols=True, max_write_samples=4096, softbit_capacity=200000)
    for s in range(0, n, 4096):
        bank.write(pcm[:, s:s + 4096])
    out = {"what": "GPU burst OQPSK bank on the whole burst recording from four starting points, against the oracle", "lib": os.path.basename(capi.LIB_PATH),
           "samples_per_channel": int(n), "seconds": n / 48000.0, "channels": []}
    tot = {"soft_bytes": 0, "soft_bytes_differing": 0, "symbols": 0, "symbols_over_1e-12": 0, "symbols_over_1e-9": 0, "symbols_over_1e-5": 0, "max_symbol_diff": 0.0, "bursts": 0}
    for c in range(nch):
        ref = O.run_burst(O.burst_oqpsk_settings(), pcm[c], chunk=4096, capture_symbols=True)
        soft, sym = bank.read_softbits(c, cap=1 << 20), bank.read_symbols(c, caprows=1 << 20)
        ok = len(soft) == len(ref["soft"]) and sym.shape == ref["symbols"].shape
        row = {"start": shifts[c], "count_ok": bool(ok)}
        if ok:
            d = np.abs(sym - ref["symbols"]).max(axis=1) if len(sym) else np.zeros(0)
            row.update({"markers_equal": bool(np.array_equal(soft == -1, ref["soft"] == -1)), "hard_decisions_equal": bool(np.array_equal(soft >= 128, ref["soft"] >= 128)),
                        "soft_bits": int(len(soft)), "soft_bytes_differing": int((soft.astype(int) != ref["soft"].astype(int)).sum()), "bursts": int((ref["soft"] == -1).sum()),
                        "symbols": int(len(d)), "max_symbol_diff": float(d.max(initial=0.0)), "symbols_over_1e-12": int((d > 1e-12).sum()),
                        "symbols_over_1e-9": int((d > 1e-9).sum()), "symbols_over_1e-5": int((d >= 1e-5).sum())})
            tot["soft_bytes"] += len(soft); tot["soft_bytes_differing"] += row["soft_bytes_differing"]; tot["symbols"] += len(d); tot["bursts"] += row["bursts"]
            for k in ("symbols_over_1e-12", "symbols_over_1e-9", "symbols_over_1e-5"):
                tot[k] += row[k]
            tot["max_symbol_diff"] = max(tot["max_symbol_diff"], row["max_symbol_diff"])
        out["channels"].append(row)
    out["total"] = tot
    bank.close()
    print(json.dumps(out))


if __name__ == "__main__":
    {"stage": stage, "cpu": cpu, "gpu": gpu}[sys.argv[1] if len(sys.argv) > 1 else "gpu"]()
