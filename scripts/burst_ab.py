#!/usr/bin/env python3
"""How close the burst demodulators are to the oracle, in numbers (one JSON line): per kind, over a bank of channels with their own bursts,
the soft bytes that differ, the largest soft-symbol difference and how many symbols differ by more than 1e-12 / 1e-9.  Run once with the
product library and once with an A/B build (JAERO_HIP_LIB) to see what an arithmetic choice buys on this path -- whose input passes an FFT
filter (the Hilbert transform) that is not the reference's FFT in the last bits whatever the kernels behind it do."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from jaero_amd import capi, signalgen as G
    from jaero_amd import demodulator as B
    from oracle import oracle as O

    capi.lib()
    out = {"lib": os.path.basename(capi.LIB_PATH)}
    rng = np.random.default_rng(11)
    for kind in ("burst_oqpsk", "burst_msk"):
        nch, n = 24, 120000
        pcm = np.zeros((nch, n), np.int16)
        for c in range(nch):
            if kind == "burst_oqpsk":
                st = [int(rng.integers(25000, 50000)), int(rng.integers(75000, 95000))]
                pcm[c], _ = G.burst_oqpsk(n, burst_starts=st, ndata_sym=700, fc=8000.0 + rng.uniform(-60, 60), ebno_db=float(rng.uniform(10, 18)), seed=G.SEED_BASE + 9500 + c)
            else:
                pcm[c], _ = G.burst_msk(n, burst_starts=[int(n * rng.uniform(0.1, 0.2)), int(n * rng.uniform(0.55, 0.65))], fb=1200.0, fc=1900.0 + rng.uniform(-300, 300),
                                        ncw=int(rng.integers(112, 148)), ebno_db=float(rng.uniform(14, 22)), seed=G.SEED_BASE + 9700 + c)
        bs = B.BurstOqpskSettings() if kind == "burst_oqpsk" else B.BurstMskSettings(freq_center=1000.0, fb=1200.0, lockingbw=1800.0)
        os_ = O.burst_oqpsk_settings() if kind == "burst_oqpsk" else O.burst_msk_settings(fb=1200.0, lockingbw=1800.0)
        bank = B.DemodulatorBank(bs, nch, capture_symbols=True, max_write_samples=4096, softbit_capacity=40000)
        for s in range(0, n, 4096):
            bank.write(pcm[:, s:s + 4096])
        r = {"channels": nch, "soft_bytes": 0, "soft_bytes_differing": 0, "hard_equal": True, "symbols": 0, "max_symbol_diff": 0.0, "symbols_over_1e-12": 0, "symbols_over_1e-9": 0,
             "bursts": 0}
        for c in range(nch):
            ref = O.run_burst(os_, pcm[c], chunk=4096, capture_symbols=True)
            soft, sym = bank.read_softbits(c), bank.read_symbols(c)
            ok = len(soft) == len(ref["soft"]) and sym.shape == ref["symbols"].shape
            if not ok:
                r["hard_equal"] = False
                continue
            r["hard_equal"] &= bool(np.array_equal(soft >= 128, ref["soft"] >= 128))
            r["soft_bytes"] += int(len(soft)); r["soft_bytes_differing"] += int((soft != ref["soft"]).sum())
            d = np.abs(sym - ref["symbols"]).max(axis=1) if len(sym) else np.zeros(0)
            r["symbols"] += int(len(d))
            if len(d) and float(d.max()) > r["max_symbol_diff"]:
                k = int(d.argmax())
                # where the worst symbol sits: its index counted from the start of its burst (soft bits come two per symbol; -1 marks a burst start)
                marks = np.flatnonzero(ref["soft"] == -1)
                r["worst"] = {"channel": c, "symbol": k, "of": int(len(d)), "gpu": [float(x) for x in sym[k]], "oracle": [float(x) for x in ref["symbols"][k]],
                              "bursts_start_at_soft_bit": [int(m) for m in marks[:4]], "neighbours_diff": [float(x) for x in d[max(0, k - 3):k + 4]]}
            r["max_symbol_diff"] = max(r["max_symbol_diff"], float(d.max(initial=0.0)))
            r["symbols_over_1e-12"] += int((d > 1e-12).sum()); r["symbols_over_1e-9"] += int((d > 1e-9).sum())
            r["bursts"] += int((ref["soft"] == -1).sum())
        bank.close()
        out[kind] = r
    print(json.dumps(out))


if __name__ == "__main__":
    main()
