#!/usr/bin/env python3
"""Fuzz the continuous Aero-L bit pipeline's device code on the CPU (tests/host_emul/aerolp_emul.cpp: k_aerol_bits<false>, k_aerol_post)
against the oracle: random frame counts, garbage prefixes, noise levels, arm inversions, erasure runs, lost and doubled stretches (short
frames, unique words out of place), write sizes from a few soft bits to several blocks.  usage: tests/fuzz/fuzz_aerolp_emul.py [rounds] [seed]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jaero_amd import aerol_frames as AF  # noqa: E402
from oracle import oracle as O  # noqa: E402  (test tool)


def build():
    O.lib()
    td = tempfile.mkdtemp(prefix="aerolp_fuzz_")
    so = os.path.join(td, "libaerolp_emul.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "tests", "host_emul", "stub"), "-o", so,
                           os.path.join(ROOT, "tests", "host_emul", "aerolp_emul.cpp"), "-L" + os.path.join(ROOT, "oracle"), "-l:liboracle.so",
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    L = C.CDLL(so)
    L.emulp_create.restype = C.c_void_p
    L.emulp_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.emulp_destroy.argtypes = [C.c_void_p]
    L.emulp_write.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.emulp_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    return L


def stream(rng, fb):
    pay = AF.random_payloads(int(rng.integers(2, 7)), fb, seed=int(rng.integers(1 << 20)))
    bits, _ = AF.p_channel_bits(pay, fb, invert_i=bool(rng.integers(2)), invert_q=bool(rng.integers(2)))
    pre = rng.integers(0, 2, size=int(rng.integers(0, 900)), dtype=np.uint8)
    soft = AF.to_soft(np.concatenate([pre, bits]), sigma=float(rng.uniform(0, 45)), seed=int(rng.integers(1 << 20))).copy()
    for _ in range(int(rng.integers(0, 4))):
        k = int(rng.integers(0, max(1, len(soft) - 100)))
        u = rng.random()
        if u < 0.35:
            soft[k:k + int(rng.integers(1, 300))] = 128
        elif u < 0.7:
            soft = np.concatenate([soft[:k], soft[k + int(rng.integers(1, 2000)):]])
        else:
            n = int(rng.integers(1, 1500))
            soft = np.concatenate([soft[:k + n], soft[k:]])
    return soft


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    L = build()
    rng = np.random.default_rng(seed)
    bad = 0
    for r in range(rounds):
        fb = int(rng.choice([600, 1200, 1200, 10500]))
        nch = int(rng.integers(1, 7))
        width = int(rng.choice([50, 333, 700, 2000, 6000, 12000]))
        streams = [stream(rng, fb) for _ in range(nch)]
        h = L.emulp_create(nch, fb, 2000)
        pos = np.zeros(nch, dtype=np.int64)
        lens = np.array([len(s) for s in streams])
        while (pos < lens).any():
            cnt = np.minimum(rng.integers(0, width + 1, size=nch), lens - pos).astype(np.int32)
            buf = np.zeros((nch, width), np.int16)
            for c in range(nch):
                buf[c, :cnt[c]] = streams[c][pos[c]:pos[c] + cnt[c]]
            L.emulp_write(h, buf.ctypes.data, cnt.ctypes.data, width, int(cnt.max()))
            pos += cnt
        for c in range(nch):
            o = O.run_aerol(fb, streams[c], 1 << 20)
            sus = np.zeros((8192, 16), np.int32)
            n = L.emulp_read(h, c, 0, sus.ctypes.data, 8192)
            ev = np.zeros((256, 3), np.int64)
            m = L.emulp_read(h, c, 1, ev.ctypes.data, 256)
            if not (np.array_equal(sus[:n], o["sus"]) and np.array_equal(ev[:m], o["events"][:256])):
                bad += 1
                np.save(f"/tmp/fuzz_aerolp_fail_{seed}_{r}_{c}.npy", streams[c])
                print(f"MISMATCH round {r} fb {fb} ch {c} width {width} (stream saved)")
        L.emulp_destroy(h)
    print(f"{rounds} rounds, seed {seed}: {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
