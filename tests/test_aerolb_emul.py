"""CPU: the device code of the R/T packet search (jaero_amd/csrc/k_aerol_burst.h: k_aerolb_bits, k_aerolb_post) compiled for the host and
run thread by thread in the rounds of jaero_aerol_write (tests/host_emul/aerolb_emul.cpp; the oracle's Decode_soft standing in for
k_viterbi) against oracle/aerol_oracle.c.  This is the multi-channel / ragged-write coverage that tests/test_gpu_aerol_burst.py asks of
the GPU, available without one -- both ways the bit walk takes its soft entries (eight at a time from 16-byte aligned rows, one at a
time otherwise) and the block words it finishes across launches."""
import ctypes as C
import importlib.util
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def E(oracle_mod):
    oracle_mod.lib()
    td = tempfile.mkdtemp(prefix="aerolb_emul_")
    so = os.path.join(td, "libaerolb_emul.so")
    cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "tests", "host_emul", "stub"), "-o", so,
           os.path.join(ROOT, "tests", "host_emul", "aerolb_emul.cpp"), "-L" + os.path.join(ROOT, "oracle"), "-l:liboracle.so",
           "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    subprocess.check_call(cmd)
    L = C.CDLL(so)
    L.emulb_create.restype = C.c_void_p
    L.emulb_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.emulb_destroy.argtypes = [C.c_void_p]
    L.emulb_write.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.emulb_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.emulb_overflow.argtypes = [C.c_void_p, C.c_int]
    L.emulb_fast_groups.restype = C.c_longlong
    L.emulb_fast_fill_groups.restype = C.c_longlong
    return L


@pytest.fixture(scope="module")
def mk():
    spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def aligned_rows(nch, width):
    """int16 [nch][width] whose base is 16-byte aligned (numpy gives 16 already on this platform; made certain here)."""
    raw = np.zeros(nch * width + 8, np.int16)
    off = (-raw.ctypes.data % 16) // 2
    return raw[off:off + nch * width].reshape(nch, width)


def run_bank(L, fb, streams, width, rng, wide):
    nch = len(streams)
    h = L.emulb_create(nch, fb, 700)
    pos = np.zeros(nch, dtype=np.int64)
    lens = np.array([len(x) for x in streams])
    buf = aligned_rows(nch, width)
    while (pos < lens).any():
        cnt = np.minimum(rng.integers(1, width + 1, size=nch), lens - pos).astype(np.int32)
        buf[:] = 0
        for c in range(nch):
            buf[c, :cnt[c]] = streams[c][pos[c]:pos[c] + cnt[c]]
        assert L.emulb_write(h, buf.ctypes.data, cnt.ctypes.data, width, int(cnt.max()), wide) == 0
        pos += cnt
    out = []
    for c in range(nch):
        rows = np.zeros((4096, 16), np.int32)
        n = L.emulb_read(h, c, 0, rows.ctypes.data, 4096)
        ev = np.zeros((256, 3), np.int64)
        m = L.emulb_read(h, c, 1, ev.ctypes.data, 256)
        assert L.emulb_overflow(h, c) == 0
        out.append((rows[:n].copy(), ev[:m].copy()))
    L.emulb_destroy(h)
    return out


@pytest.mark.parametrize("width,wide", [(3000, 1), (3000, 0), (2996, -1), (200, 1)])
def test_oqpsk_bank_logic_vs_oracle(E, oracle_mod, mk, width, wide):
    """The cases of tests/test_gpu_aerol_burst.py::test_bank_vs_oracle (fewer channels): different packets, noise levels, arm inversions,
    lost tails / late unique words, noise-only channels, ragged writes.  width 200: many launches per packet, so that block words are
    begun in one launch and finished in a later one at every byte offset."""
    nch = 24
    rng = np.random.default_rng(3)
    streams = []
    for c in range(nch):
        if c % 7 == 6:
            streams.append(np.clip(np.round(128 + rng.normal(0, 40, 12000)), 0, 255).astype(np.int16))
        else:
            _, x = mk.rt_case(100 + c, float(rng.uniform(8, 45)), (bool(c & 1), bool(c & 2)), cut=(c % 3 == 0))
            streams.append(x)
    fast0, fill0 = E.emulb_fast_groups(), E.emulb_fast_fill_groups()
    got = run_bank(E, 10500, streams, width, rng, wide)
    fast, fill = E.emulb_fast_groups() - fast0, E.emulb_fast_fill_groups() - fill0
    # inside and behind a packet the walk takes aligned groups of eight inert entries in one go (k_aerolb_bits<true>); never from unaligned rows
    assert (fast > 200 and fill > 200) if wide == 1 else (fast == 0), (fast, fill)
    npk = 0
    for c in range(nch):
        o = oracle_mod.run_aerol_burst(10500, streams[c])
        want = oracle_mod.packets_from_rows(o["packets"])
        assert oracle_mod.packets_from_rows(got[c][0]) == want, c
        assert np.array_equal(got[c][1], o["events"]), c
        npk += len(want)
    assert npk > 40


@pytest.mark.parametrize("fb,width,wide", [(1200, 2504, 1), (600, 2500, -1)])
def test_msk_bank_logic_vs_oracle(E, oracle_mod, mk, fb, width, wide):
    nch = 22
    rng = np.random.default_rng(fb)
    streams = []
    for c in range(nch):
        if c % 11 == 10:
            streams.append(np.clip(np.round(128 + rng.normal(0, 40, 9000)), 0, 255).astype(np.int16))
        else:
            _, x = mk.rt_case_msk(200 + c, float(rng.uniform(8, 40)), invert=bool(c & 1), cut=(c % 3 == 0))
            streams.append(x)
    got = run_bank(E, fb, streams, width, rng, wide)
    npk = 0
    for c in range(nch):
        o = oracle_mod.run_aerol_burst(fb, streams[c])
        want = oracle_mod.packets_from_rows(o["packets"])
        assert oracle_mod.packets_from_rows(got[c][0]) == want, c
        assert np.array_equal(got[c][1], o["events"]), c
        npk += len(want)
    assert npk > 40


def short_gap_streams(nch, seed0=900):
    """Bursts that follow each other while the frame countdown of the one before still runs (gap 1500 .. 3300 entries, every residue
    modulo 8): the next start-of-burst marker arrives in the stretch the walk takes eight entries at a time."""
    from jaero_amd import aerol_frames as AF

    streams = []
    for c in range(nch):
        rng = np.random.default_rng(seed0 + c)
        rb = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))
        pk = [("R", rb(17)), ("T", (rb(4), [rb(10) for _ in range(2 + c % 5)])), ("R", rb(17)), ("T", (rb(4), [rb(10) for _ in range(3)])), ("R", rb(17))]
        x = AF.rt_burst_stream(pk, sigma=14.0 + c, seed=seed0 + c, invert_i=bool(c & 1), invert_q=bool(c & 2), gap=1500 + 113 * c, lead=80 + 2 * (c % 4))
        streams.append(x[(c % 8):])  # the markers at every position of an aligned group
    return streams


@pytest.mark.parametrize("width,wide", [(4096, 1), (4096, 0), (520, 1)])
def test_oqpsk_short_gaps_vs_oracle(E, oracle_mod, width, wide):
    streams = short_gap_streams(16)
    rng = np.random.default_rng(11)
    fast0, fill0 = E.emulb_fast_groups(), E.emulb_fast_fill_groups()
    got = run_bank(E, 10500, streams, width, rng, wide)
    fast, fill = E.emulb_fast_groups() - fast0, E.emulb_fast_fill_groups() - fill0
    assert (fast > 1000 and fill > 500) if wide == 1 else (fast == 0), (fast, fill)
    npk = 0
    for c in range(len(streams)):
        o = oracle_mod.run_aerol_burst(10500, streams[c])
        want = oracle_mod.packets_from_rows(o["packets"])
        assert oracle_mod.packets_from_rows(got[c][0]) == want, c
        assert np.array_equal(got[c][1], o["events"]), c
        npk += len(want)
    assert npk >= 2 * len(streams)  # with gaps this short not every burst is found (the countdown of the one before still runs): on both sides
