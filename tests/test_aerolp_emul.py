"""CPU: the device code of the continuous Aero-L bit pipeline (jaero_amd/csrc/k_aerol.h: k_aerol_bits<false>, k_aerol_post, aerol_frame_end)
compiled for the host and run thread by thread in the rounds of jaero_aerol_write (tests/host_emul/aerolp_emul.cpp; the oracle's
Decode_Continuous standing in for k_viterbi) against oracle/aerol_oracle.c: the multi-channel / ragged-write coverage of
tests/test_gpu_aerol.py::test_bank_vs_oracle without a GPU.  (10 500 bps runs the bit-by-bit variant here; the jumping one needs the
wavefront kernels k_aerol_scan / k_aerol_bulk and is the GPU tests'.)"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from jaero_amd import aerol_frames as AF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def E(oracle_mod):
    oracle_mod.lib()
    td = tempfile.mkdtemp(prefix="aerolp_emul_")
    so = os.path.join(td, "libaerolp_emul.so")
    cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "tests", "host_emul", "stub"), "-o", so,
           os.path.join(ROOT, "tests", "host_emul", "aerolp_emul.cpp"), "-L" + os.path.join(ROOT, "oracle"), "-l:liboracle.so",
           "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    subprocess.check_call(cmd)
    L = C.CDLL(so)
    L.emulp_create.restype = C.c_void_p
    L.emulp_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.emulp_destroy.argtypes = [C.c_void_p]
    L.emulp_write.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.emulp_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.emulp_overflow.argtypes = [C.c_void_p, C.c_int]
    return L


@pytest.mark.parametrize("fb,nch,width", [(1200, 70, 6000), (600, 20, 700), (10500, 12, 6000)])
def test_bank_logic_vs_oracle(E, oracle_mod, fb, nch, width):
    """Different frames / noise / arm inversions / garbage prefixes per channel, ragged per-channel counts in every write (width 700: less
    than a block per write, so blocks complete across writes)."""
    rng = np.random.default_rng(fb)
    streams = []
    for c in range(nch):
        pay = AF.random_payloads(5, fb, seed=1000 + c)
        bits, _ = AF.p_channel_bits(pay, fb, invert_i=bool(c & 1), invert_q=bool(c & 2))
        pre = rng.integers(0, 2, size=int(rng.integers(0, 900)), dtype=np.uint8)
        streams.append(AF.to_soft(np.concatenate([pre, bits]), sigma=float(rng.uniform(0, 45)), seed=c))
    h = E.emulp_create(nch, fb, 400)
    pos = np.zeros(nch, dtype=np.int64)
    lens = np.array([len(s) for s in streams])
    while (pos < lens).any():
        cnt = np.minimum(rng.integers(1, width, size=nch), lens - pos).astype(np.int32)
        buf = np.zeros((nch, width), np.int16)
        for c in range(nch):
            buf[c, :cnt[c]] = streams[c][pos[c]:pos[c] + cnt[c]]
        assert E.emulp_write(h, buf.ctypes.data, cnt.ctypes.data, width, int(cnt.max())) == 0
        pos += cnt
    nclean = 0
    for c in range(nch):
        o = oracle_mod.run_aerol(fb, streams[c], 1 << 20)
        sus = np.zeros((4096, 16), np.int32)
        n = E.emulp_read(h, c, 0, sus.ctypes.data, 4096)
        ev = np.zeros((256, 3), np.int64)
        m = E.emulp_read(h, c, 1, ev.ctypes.data, 256)
        assert E.emulp_overflow(h, c) == 0
        assert np.array_equal(sus[:n], o["sus"]), c
        assert np.array_equal(ev[:m], o["events"]), c
        nclean += int(o["sus"][:, 14].sum())
    assert nclean > nch * 10
    E.emulp_destroy(h)
