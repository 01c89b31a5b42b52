#!/usr/bin/env python3
"""Fuzz the Aero-L C-channel pipeline's device code on the CPU (tests/host_emul/aerolc_emul.cpp) against the oracle: random frame counts,
leads, noise levels, arm inversions, copies of the unique word planted anywhere (inside and outside the detection windows), erasure runs,
lost and doubled stretches (frames that come out short or long), write sizes from a few soft bits to several frames.
usage: tests/fuzz/fuzz_aerolc_emul.py [rounds] [seed]"""
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
    td = tempfile.mkdtemp(prefix="aerolc_fuzz_")
    so = os.path.join(td, "libaerolc_emul.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "host_emul", "aerolc_emul.cpp"),
                           "-L" + os.path.join(ROOT, "oracle"), "-l:liboracle.so", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    L = C.CDLL(so)
    L.emul_create.restype = C.c_void_p
    L.emul_create.argtypes = [C.c_int, C.c_int]
    L.emul_destroy.argtypes = [C.c_void_p]
    L.emul_write.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.emul_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    return L


def drain(L, h, ch, which, rowshape, dtype, cap=8192):
    buf = np.zeros((cap,) + rowshape, dtype=dtype)
    n = L.emul_read(h, ch, which, buf.ctypes.data, cap)
    return buf[:n]


def oracle_run(soft, group=32):
    a = O.AeroL(8400)
    for s in range(0, len(soft), group):
        a.write(soft[s:s + group])
    fn, voice = a.take_voice()
    return fn, voice, a.take_sus(), a.take_events()


def stream(rng, c):
    lead = int(rng.integers(0, 4200))
    _, soft = AF.c_channel_case(int(rng.integers(1 << 20)), int(rng.integers(2, 7)), float(rng.uniform(5, 45)), inv=(bool(rng.integers(2)), bool(rng.integers(2))), lead=lead)
    soft = soft.copy()
    uw = soft[lead:lead + 104].copy()
    for _ in range(int(rng.integers(0, 5))):
        k = int(rng.integers(0, max(1, len(soft) - 200)))
        u = rng.random()
        if u < 0.45:  # a copy of the unique word anywhere (either arm parity)
            soft[k:k + 104] = uw[:len(soft[k:k + 104])]
        elif u < 0.6:
            soft[k:k + int(rng.integers(1, 600))] = 128
        elif u < 0.8:  # lost stretch
            soft = np.concatenate([soft[:k], soft[k + int(rng.integers(1, 5000)):]])
        else:  # doubled stretch
            n = int(rng.integers(1, 3000))
            soft = np.concatenate([soft[:k + n], soft[k:]])
    return soft


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    L = build()
    rng = np.random.default_rng(seed)
    bad = 0
    for r in range(rounds):
        nch = int(rng.integers(1, 7))
        write = int(rng.choice([40, 300, 3000, 4200, 5000, 9000, 17000]))
        streams = [stream(rng, c) for c in range(nch)]
        h = L.emul_create(nch, 600)
        pos = [0] * nch
        while any(pos[c] < len(streams[c]) for c in range(nch)):
            cnt = np.array([min(int(rng.integers(0, write + 1)), len(streams[c]) - pos[c]) for c in range(nch)], dtype=np.int32)
            buf = np.zeros((nch, write), dtype=np.int16)
            for c in range(nch):
                buf[c, :cnt[c]] = streams[c][pos[c]:pos[c] + cnt[c]]
                pos[c] += int(cnt[c])
            L.emul_write(h, buf.ctypes.data, cnt.ctypes.data, write, int(cnt.max()))
        for c in range(nch):
            ofn, ovoice, osus, oev = oracle_run(streams[c])
            v = drain(L, h, c, 1, (304,), np.uint8)
            fn = v[:, :4].copy().view(np.uint32).reshape(-1)
            ok = (np.array_equal(fn, ofn) and np.array_equal(v[:, 4:], ovoice) and np.array_equal(drain(L, h, c, 0, (16,), np.int32), osus)
                  and np.array_equal(drain(L, h, c, 2, (3,), np.int64), oev[:256]))
            if not ok:
                bad += 1
                np.save(f"/tmp/fuzz_aerolc_fail_{seed}_{r}_{c}.npy", streams[c])
                print(f"MISMATCH round {r} ch {c} write {write} (stream saved)")
        L.emul_destroy(h)
    print(f"{rounds} rounds, seed {seed}: {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
