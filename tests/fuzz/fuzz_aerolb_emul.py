#!/usr/bin/env python3
"""Fuzz the R/T packet search's device code on the CPU (tests/host_emul/aerolb_emul.cpp) against the oracle: random packets, gaps (short
ones too: the next burst inside the countdown of the one before), noise levels, arm inversions, stray start-of-burst markers, runs of
erasures (128), lost stretches, write sizes and row widths (aligned and not).  usage: tests/fuzz/fuzz_aerolb_emul.py [rounds] [seed]"""
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
    td = tempfile.mkdtemp(prefix="aerolb_fuzz_")
    so = os.path.join(td, "libaerolb_emul.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "tests", "host_emul", "stub"), "-o", so,
                           os.path.join(ROOT, "tests", "host_emul", "aerolb_emul.cpp"), "-L" + os.path.join(ROOT, "oracle"), "-l:liboracle.so",
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    L = C.CDLL(so)
    L.emulb_create.restype = C.c_void_p
    L.emulb_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.emulb_destroy.argtypes = [C.c_void_p]
    L.emulb_write.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.emulb_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.emulb_fast_groups.restype = C.c_longlong
    return L


def stream(rng, fb):
    rb = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))
    pk = []
    for _ in range(int(rng.integers(2, 6))):
        if rng.random() < 0.4:
            pk.append(("R", rb(17)))
        else:
            nsu = int(rng.integers(2, 12)) if fb == 10500 else int(rng.choice([4, 9, 15, 16]))
            pk.append(("T", (rb(4), [rb(10) for _ in range(nsu)])))
    gap = int(rng.choice([rng.integers(200, 3000), rng.integers(3000, 12000)]))
    sigma = float(rng.uniform(5, 45))
    seed = int(rng.integers(1 << 30))
    lead = int(rng.integers(40, 200))
    if fb == 10500:
        x = AF.rt_burst_stream(pk, sigma=sigma, seed=seed, invert_i=bool(rng.integers(2)), invert_q=bool(rng.integers(2)), gap=gap, lead=lead)
    else:
        x = AF.rt_burst_stream_msk(pk, sigma=sigma, seed=seed, invert=bool(rng.integers(2)), gap=gap, lead=lead)
    x = x[int(rng.integers(0, 16)):].copy()
    for _ in range(int(rng.integers(0, 4))):  # stray markers, erasure runs, lost stretches
        k = int(rng.integers(0, len(x)))
        u = rng.random()
        if u < 0.4:
            x[k] = -1
        elif u < 0.7:
            x[k:k + int(rng.integers(1, 400))] = 128
        else:
            x = np.concatenate([x[:k], x[k + int(rng.integers(1, 900)):]])
    return legalise(x, rng)


def legalise(x, rng):
    """A burst demodulator emits soft bits in pairs and a start-of-burst marker only between pairs (burstoqpskdemodulator.cpp:546-585); the
    grouping both sides re-derive is defined for such streams only.  One soft entry is put in front of every marker that would split a pair."""
    out, even = [], True
    for v in x.tolist():
        if v < 0:
            if not even:
                out.append(int(np.clip(round(128 + rng.normal(0, 40)), 0, 255)))
                even = True
            out.append(v)
        else:
            out.append(v)
            even = not even
    return np.array(out, dtype=np.int16)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    L = build()
    rng = np.random.default_rng(seed)
    bad = 0
    for r in range(rounds):
        fb = int(rng.choice([10500, 10500, 10500, 1200, 600]))
        nch = int(rng.integers(1, 9))
        width = int(rng.choice([64, 200, 520, 1000, 3000, 4096, 2996, 777]))
        wide = -1 if width % 8 else int(rng.choice([1, 1, 0]))
        streams = [stream(rng, fb) for _ in range(nch)]
        h = L.emulb_create(nch, fb, 2000)
        pos = np.zeros(nch, dtype=np.int64)
        lens = np.array([len(x) for x in streams])
        raw = np.zeros(nch * width + 8, np.int16)
        off = (-raw.ctypes.data % 16) // 2
        buf = raw[off:off + nch * width].reshape(nch, width)
        while (pos < lens).any():
            cnt = np.minimum(rng.integers(0, width + 1, size=nch), lens - pos).astype(np.int32)
            buf[:] = 0
            for c in range(nch):
                buf[c, :cnt[c]] = streams[c][pos[c]:pos[c] + cnt[c]]
            assert L.emulb_write(h, buf.ctypes.data, cnt.ctypes.data, width, int(cnt.max()), wide) == 0
            pos += cnt
        for c in range(nch):
            rows = np.zeros((8192, 16), np.int32)
            n = L.emulb_read(h, c, 0, rows.ctypes.data, 8192)
            ev = np.zeros((256, 3), np.int64)
            m = L.emulb_read(h, c, 1, ev.ctypes.data, 256)
            o = O.run_aerol_burst(fb, streams[c])
            ok = O.packets_from_rows(rows[:n]) == O.packets_from_rows(o["packets"]) and np.array_equal(ev[:m], o["events"][:256])
            if not ok:
                bad += 1
                np.save(f"/tmp/fuzz_aerolb_fail_{seed}_{r}_{c}.npy", streams[c])
                print(f"MISMATCH round {r} fb {fb} ch {c} width {width} wide {wide} (stream saved)")
        L.emulb_destroy(h)
    print(f"{rounds} rounds, seed {seed}: {bad} mismatches, {L.emulb_fast_groups()} groups of eight taken at once")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
