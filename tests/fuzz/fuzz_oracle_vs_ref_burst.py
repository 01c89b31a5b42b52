#!/usr/bin/env python3
"""Fuzz the oracle's burst-demodulator restatement (oracle/jaero_oracle_burst.c) against the UNMODIFIED BurstOqpskDemodulator /
BurstMskDemodulator (oracle/_ref): random burst positions and counts, carrier offsets, Eb/N0, write sizes, AFC, and CenterFreqChangedSlot at
a random moment.  Soft bits (with the -1 markers) and every emission must be identical.  Needs /root/reference.
usage: tests/fuzz/fuzz_oracle_vs_ref_burst.py [rounds] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jaero_amd import signalgen as G  # noqa: E402
from oracle import oracle as O  # noqa: E402  (test tool)


def as_write_stamps(events, chunk):
    ev = events[events[:, 1] < 3].copy()
    ev[:, 0] = np.floor(ev[:, 0] / chunk) * chunk
    return ev


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    assert O.have_ref(), "oracle/_ref missing: make -C oracle ref"
    O.lib()
    rng = np.random.default_rng(seed)
    bad = nb = 0
    for r in range(rounds):
        kind = str(rng.choice(["burstoqpsk", "burstmsk1200", "burstmsk600"]))
        chunk = int(rng.choice([1000, 1500, 2000, 3000, 4096, 5000]))
        afc = int(rng.integers(2))
        sd = int(rng.integers(1 << 20))
        kv, okw = {}, {}
        if kind == "burstoqpsk":
            n = int(rng.integers(120000, 260000))
            starts = sorted(int(v) for v in rng.integers(20000, n - 40000, size=int(rng.integers(1, 4))))
            starts = [s for k, s in enumerate(starts) if k == 0 or s - starts[k - 1] > 30000]
            pcm, _ = G.burst_oqpsk(n, burst_starts=starts, ndata_sym=int(rng.integers(400, 1600)), fc=8000.0 + float(rng.uniform(-40, 40)),
                                   ebno_db=float(rng.uniform(10, 18)), seed=sd)
            st, rk = O.burst_oqpsk_settings(), "burstoqpsk"
        else:
            fb = 1200.0 if kind == "burstmsk1200" else 600.0
            n = int(48000 * rng.uniform(3.0, 5.0) * (1200 / fb))
            starts = sorted(int(v) for v in rng.integers(20000, n - int(90000 * 1200 / fb), size=int(rng.integers(1, 3))))
            starts = [s for k, s in enumerate(starts) if k == 0 or s - starts[k - 1] > int(100000 * 1200 / fb)]
            pcm, _ = G.burst_msk(n, burst_starts=starts, fb=fb, fc=1900.0 + float(rng.uniform(-30, 30)), ebno_db=float(rng.uniform(14, 20)), seed=sd)
            st, rk = O.burst_msk_settings(fb=fb, lockingbw=1.5 * fb), "burstmsk"
            kv.update(fb=int(fb), lockingbw=1.5 * fb)
            if rng.random() < 0.4:
                t = int(rng.integers(1000, n)); hz = float(rng.choice([100.0, 1400.0, 1800.0, 2300.0]))
                kv.update(center_at=t, center_hz=hz); okw.update(center_at=t, center_hz=hz)
        ref = O.run_ref(rk, pcm, chunk=chunk, afc=afc, **kv)
        got = O.run_burst(st, pcm, chunk=chunk, afc=bool(afc), **okw)
        nb += int((ref["soft"] == -1).sum())
        ok = np.array_equal(ref["soft"], got["soft"]) and np.array_equal(ref["events"], as_write_stamps(got["events"], chunk))
        if not ok:
            bad += 1
            print(f"MISMATCH round {r}: {kind} n={n} seed={sd} chunk={chunk} afc={afc} starts={starts} kv={kv}")
    print(f"{rounds} rounds, seed {seed}: {bad} mismatches, {nb} bursts accepted by the reference")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
