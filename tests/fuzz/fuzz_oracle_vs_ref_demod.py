#!/usr/bin/env python3
"""Fuzz the oracle's demodulator restatement (oracle/jaero_oracle.c) against the UNMODIFIED reference demodulators (oracle/_ref) on random
cases: kind and rate (OQPSK 10 500 / 8400, MSK 600 / 1200), carrier offset, Eb/N0, write size, initial AFC / SQL / cpuReduce, and a random subset
of the slots a running object can receive between two writes -- DCDstatSlot on and off, CenterFreqChangedSlot, setSettings (centre frequency and
locking bandwidth), two flag changes -- at random moments.  Soft bits and status rows must be identical.  Needs /root/reference.
usage: tests/fuzz/fuzz_oracle_vs_ref_demod.py [rounds] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jaero_amd import signalgen as G  # noqa: E402
from oracle import oracle as O  # noqa: E402  (test tool)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    assert O.have_ref(), "oracle/_ref missing: make -C oracle ref"
    O.lib()
    rng = np.random.default_rng(seed)
    bad = 0
    for r in range(rounds):
        kind = str(rng.choice(["oqpsk", "oqpsk8400", "msk1200", "msk600"]))
        n = int(rng.integers(90000, 200000))
        sd = int(rng.integers(1 << 20))
        chunk = int(rng.choice([512, 1000, 1500, 2500, 4096, 5000, 8192]))
        f0 = [int(rng.integers(2)) for _ in range(3)]
        if kind.startswith("oqpsk"):
            fb = 10500.0 if kind == "oqpsk" else 8400.0
            fc0, bw = 8000.0, fb
            pcm, _ = G.oqpsk(n, fb=fb, fc=fc0 + float(rng.uniform(-60, 60)), ebno_db=float(rng.uniform(6, 14)), seed=sd)
            st = O.oqpsk_settings(fb=fb, lockingbw=bw)
            rk, kv = "oqpsk", dict(fb=int(fb), lockingbw=int(bw))
            mk = lambda fc, lbw: O.oqpsk_settings(fb=fb, freq_center=fc, lockingbw=lbw)
        else:
            fb = 1200.0 if kind == "msk1200" else 600.0
            fc0, bw = 1000.0, 1800.0 if fb == 1200.0 else 900.0
            pcm, _ = G.msk(n, fb=fb, fc=fc0 + float(rng.uniform(-25, 25)), ebno_db=float(rng.uniform(8, 14)), seed=sd)
            st = O.msk_settings(fb=fb, lockingbw=bw)
            rk, kv = "msk", dict(fb=int(fb), lockingbw=int(bw))
            mk = lambda fc, lbw: O.msk_settings(fb=fb, freq_center=fc, lockingbw=lbw)
        okw = {}
        if rng.random() < 0.5:
            t = int(rng.integers(1000, n))
            kv["dcd_at"] = t; okw["dcd_at"] = t
            if rng.random() < 0.5:
                t2 = int(rng.integers(t + 1, n + 1))
                kv["dcd_off_at"] = t2; okw["dcd_off_at"] = t2
        if rng.random() < 0.35:
            t = int(rng.integers(1000, n))
            hz = fc0 + float(rng.integers(-80, 81)) * (1.0 if kind.startswith("oqpsk") else 0.3)
            kv["center_at"] = t; kv["center_hz"] = hz; okw["center_at"] = t; okw["center_hz"] = hz
        if rng.random() < 0.35:
            t = int(rng.integers(1000, n))
            nfc = fc0 + float(rng.integers(-40, 41)) * (1.0 if kind.startswith("oqpsk") else 0.4)
            nbw = float(rng.choice([bw, bw * 0.8 if kind.startswith("oqpsk") else bw * 0.9]))
            kv["set_at"] = t; kv["set_freq_center"] = nfc; kv["set_lockingbw"] = nbw
            okw["set_at"] = t; okw["set_settings"] = mk(nfc, nbw)
        ev = []
        if rng.random() < 0.5:
            t = int(rng.integers(1000, n)); f = [int(rng.integers(2)) for _ in range(3)]
            kv.update(flags_at=t, flags_afc=f[0], flags_sql=f[1], flags_cpureduce=f[2]); ev.append((t, bool(f[0]), bool(f[1]), bool(f[2])))
            if rng.random() < 0.5:
                t2 = int(rng.integers(t, n)); f = [int(rng.integers(2)) for _ in range(3)]
                kv.update(flags_at2=t2, flags2_afc=f[0], flags2_sql=f[1], flags2_cpureduce=f[2]); ev.append((t2, bool(f[0]), bool(f[1]), bool(f[2])))
        ref = O.run_ref(rk, pcm, afc=f0[0], sql=f0[1], cpureduce=f0[2], chunk=chunk, **kv)
        got = O.run_demod(st, pcm, afc=bool(f0[0]), sql=bool(f0[1]), cpu_reduce=bool(f0[2]), chunk=chunk, flags_events=ev, **okw)
        ok = (np.array_equal(ref["soft"], got["soft"]) and ref["status"].shape == got["status"].shape
              and np.array_equal(ref["status"][:, [0, 1, 2, 3, 5]], got["status"][:, [0, 1, 2, 3, 5]]))
        if not ok:
            bad += 1
            print(f"MISMATCH round {r}: {kind} n={n} seed={sd} chunk={chunk} f0={f0} kv={kv}")
    print(f"{rounds} rounds, seed {seed}: {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
