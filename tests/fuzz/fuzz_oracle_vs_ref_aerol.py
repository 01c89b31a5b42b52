#!/usr/bin/env python3
"""Fuzz the oracle's Aero-L restatement (oracle/aerol_oracle.c) against the UNMODIFIED AeroL (oracle/_ref) on the streams of the three
emulation fuzzers (tests/fuzz/fuzz_aerol{p,b,c}_emul.py: random frames / packets, planted unique words, markers, erasures, lost and doubled
stretches): what the reference prints -- signal units with their CRC verdicts, R / T packets and ' Bad R/T Packet' notices, voice frames and
sub-band units -- against the oracle's rows.  Needs /root/reference (to build oracle/_ref).  usage: fuzz_oracle_vs_ref_aerol.py [rounds] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_aerolb_emul as FB  # noqa: E402
import fuzz_aerolc_emul as FC  # noqa: E402
import fuzz_aerolp_emul as FP  # noqa: E402
from oracle import oracle as O  # noqa: E402  (test tool)
from test_aerol_oracle import burst_rows, c_run, c_voice_equal  # noqa: E402


def ref_burst_rows(ref):
    want = []
    for p in ref:
        if p[0] == "R":
            want.append([1, 17, 0] + list(p[1]) + [0] * (10 * 31 + 4 - 17))
        else:
            flat = [v for su in p[3] for v in su]
            want.append([2, len(p[3]), p[2]] + list(p[1]) + flat + [0] * (10 * 31 - len(flat)))
    return np.array(want, dtype=np.int32).reshape(-1, 317)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    assert O.have_ref(), "oracle/_ref missing: make -C oracle ref"
    O.lib()
    rng = np.random.default_rng(seed)
    bad = 0
    n = {"p": 0, "b": 0, "c": 0}
    for r in range(rounds):
        kind = "pbc"[r % 3]
        n[kind] += 1
        if kind == "p":
            fb = int(rng.choice([600, 1200, 10500]))
            x = FP.stream(rng, fb)
            grp = 32 if fb == 10500 else 12
            ref, _ = O.run_ref_aerol(fb, x, grp)
            o = O.run_aerol(fb, x, grp)
            mine = [(int(q[1]), bytes(q[2:12].astype(np.uint8)), bool(q[14])) for q in o["sus"]]
            ok = ref == mine
        elif kind == "b":
            fb = int(rng.choice([10500, 10500, 1200, 600]))
            x = FB.stream(rng, fb)
            ref, nbad, _ = O.run_ref_aerol_burst(fb, x)
            o = O.run_aerol_burst(fb, x)
            got = burst_rows(O.packets_from_rows(o["packets"]), msk=fb != 10500)
            ok = np.array_equal(got, ref_burst_rows(ref)) and int((o["events"][:, 1] == 3).sum()) == nbad
        else:
            x = FC.stream(rng, 0)
            voice_ref, sus_ref, _, _ = O.run_ref_aerol_c(x, 32)
            fn, voice, sus, printed, _ = c_run(O, x)
            ok = c_voice_equal(voice_ref, voice) and printed == sus_ref
        if not ok:
            bad += 1
            np.save(f"/tmp/fuzz_oracle_ref_fail_{seed}_{r}_{kind}.npy", x)
            print(f"MISMATCH round {r} kind {kind} (stream saved)")
    print(f"{rounds} rounds, seed {seed}: {bad} mismatches ({n})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
