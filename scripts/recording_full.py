#!/usr/bin/env python3
"""The WHOLE of the reference's 10.5 kbps sample recording (samples/10.5k_sample.ogg, four minutes) through reference, oracle and GPU.

One-off validation, not a test: the decoded PCM (23 MB) is not committed.  Stage it with
    python scripts/recording_full.py stage            (build container: decodes with scripts/vorbis_decode.py, resamples to 48 kHz)
then
    python scripts/recording_full.py cpu              (build container: unmodified reference against the oracle, soft bits / status / AeroL)
    python scripts/recording_full.py gpu              (GPU box: a bank fed the recording from five starting points against the oracle, and
                                                       PCM -> demodulator bank -> Aero-L bank on the device against the oracle's chain)
Each prints one JSON line (kept under profiles/)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAGE = os.path.join(ROOT, "gpurun_stage", "recording_oqpsk_10k5_48k.i16")


def stage():
    from scipy.signal import resample_poly
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import vorbis_decode

    x, rate = vorbis_decode.decode("/root/reference/samples/10.5k_sample.ogg")
    y = resample_poly(x[0], 160, 147)
    pcm = np.clip(np.round(y * 32767.0), -32768, 32767).astype(np.int16)
    os.makedirs(os.path.dirname(STAGE), exist_ok=True)
    pcm.tofile(STAGE)
    print(json.dumps({"staged": STAGE, "samples": int(len(pcm)), "seconds": len(pcm) / 48000.0, "source_rate": rate}))


def cpu():
    from oracle import oracle as O

    pcm = np.fromfile(STAGE, dtype=np.int16)
    r = O.run_ref("oqpsk", pcm)
    o = O.run_demod(O.oqpsk_settings(), pcm, chunk=4096)
    sus_ref, _ = O.run_ref_aerol(10500, r["soft"], 32)
    a = O.run_aerol(10500, r["soft"])["sus"]
    rows = np.array([[k] + list(b) + [int(ok)] for k, b, ok in sus_ref], dtype=np.int16).reshape(-1, 12)
    same_sus = a.shape[0] == rows.shape[0] and np.array_equal(a[:, 1], rows[:, 0]) and np.array_equal(a[:, 2:12], rows[:, 1:11]) and np.array_equal(a[:, 14], rows[:, 11])
    print(json.dumps({"what": "unmodified reference against the restatement on the whole recording", "samples": int(len(pcm)), "seconds": len(pcm) / 48000.0,
                      "soft_bits": int(len(r["soft"])), "soft_bits_identical": bool(np.array_equal(r["soft"], o["soft"])),
                      "status_rows": int(r["status"].shape[0]), "status_rows_identical": bool(r["status"].shape == o["status"].shape and np.array_equal(r["status"], o["status"])),
                      "carrier_hz_last": float(r["status"][-1, 1]), "ebno_db_last": float(r["status"][-1, 4]),
                      "signal_units_printed": int(rows.shape[0]), "crc_clean": int(rows[:, 11].sum()), "signal_units_identical": bool(same_sus)}))


def gpu():
    from jaero_amd import capi
    from jaero_amd import demodulator as B
    from oracle import oracle as O

    capi.lib()
    full = np.fromfile(STAGE, dtype=np.int16)
    shifts = [0, 1234, 7777, 20001, 48000]  # four + the one that used to leave the tolerance (round 4: symbol 25 282)
    n = (len(full) - max(shifts)) // 24000 * 24000
    pcm = np.stack([full[s:s + n] for s in shifts])
    nch = len(shifts)
    demod = B.DemodulatorBank(B.OqpskSettings(), nch, device=0, ebno=True, status_log=True, capture_symbols=True, max_write_samples=24000,
                              softbit_capacity=2 * n * 10500 // 48000 + 4096)
    for s in range(0, n, 24000):
        demod.write(pcm[:, s:s + 24000])
    out = {"what": "GPU bank on the whole recording from five starting points, against the oracle", "samples_per_channel": int(n), "seconds": n / 48000.0, "channels": []}
    for c in range(nch):
        ref = O.run_demod(O.oqpsk_settings(), pcm[c], chunk=24000, capture_symbols=True)
        soft, sym, log = demod.read_softbits(c, cap=1 << 22), demod.read_symbols(c, caprows=1 << 21), demod.read_status_log(c, caprows=1 << 13)
        m = len(ref["soft"])
        d = np.abs(sym - ref["symbols"]).max(axis=1) if sym.shape == ref["symbols"].shape else np.array([np.inf])
        out["channels"].append({"start": shifts[c], "soft_bits": int(m), "count_ok": bool(len(soft) == m + ref["pending"]),
                                "hard_decisions_equal": bool(np.array_equal(soft[:m] >= 128, ref["soft"] >= 128)),
                                "max_soft_byte_diff": int(np.max(np.abs(soft[:m].astype(int) - ref["soft"].astype(int)), initial=0)),
                                "soft_bytes_differing": int((soft[:m].astype(int) != ref["soft"].astype(int)).sum()),
                                "symbols": int(len(d)), "max_symbol_diff": float(d.max()), "symbols_over_1e-5": int((d >= 1e-5).sum()),
                                "symbols_over_1e-5_per_1e6": float((d >= 1e-5).sum()) * 1e6 / max(1, len(d)),
                                "status_rows": int(log.shape[0]), "status_rows_ok": bool(log.shape == ref["status"].shape and np.array_equal(log[:, [0, 5]], ref["status"][:, [0, 5]])),
                                "max_status_diff": float(np.max(np.abs(log[:, 1:5] - ref["status"][:, 1:5]))) if log.shape == ref["status"].shape else None})
    demod.close()
    # the device chain, half a second per write, against the oracle's chain written the same way
    demod = B.DemodulatorBank(B.OqpskSettings(), nch, device=0, max_write_samples=24000, softbit_capacity=8192)
    aerol = B.AeroLBank(nch, 10500, max_softbits_per_write=8192, su_capacity=13000)
    for s in range(0, n, 24000):
        demod.write(pcm[:, s:s + 24000])
        aerol.write_from_bank(demod, 8192)
    for c in range(nch):
        sus = aerol.read_sus(c, caprows=13000)
        d, al = O.Demod(O.oqpsk_settings()), O.AeroL(10500)
        for s in range(0, n, 24000):
            d.write(pcm[c, s:s + 24000])
            al.write(d.take_soft())
        osus = al.take_sus()
        out["channels"][c].update({"signal_units_printed": int(sus.shape[0]), "crc_clean": int(sus[:, 14].sum()),
                                   "signal_units_identical_to_oracle_chain": bool(sus.shape == osus.shape and np.array_equal(sus[:, 1:15], osus[:, 1:15]))})
    demod.close()
    aerol.close()
    out["symbols_total"] = int(sum(c["symbols"] for c in out["channels"]))
    out["symbols_over_1e-5_total"] = int(sum(c["symbols_over_1e-5"] for c in out["channels"]))
    out["symbols_over_1e-5_per_1e6"] = out["symbols_over_1e-5_total"] * 1e6 / max(1, out["symbols_total"])
    print(json.dumps(out))


if __name__ == "__main__":
    {"stage": stage, "cpu": cpu, "gpu": gpu}[sys.argv[1]]()
