"""CPU: the device code of the Aero-L C-channel pipeline (jaero_amd/csrc/aerolc.h) compiled for the host and run thread by thread
(tests/host_emul/aerolc_emul.cpp, the oracle's Viterbi standing in for k_viterbi) against oracle/aerol_oracle.c.  This is the
multi-channel / ragged-write coverage that tests/test_gpu_aerol_c.py::test_bank_vs_oracle asks of the GPU, available without one."""
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
    td = tempfile.mkdtemp(prefix="aerolc_emul_")
    so = os.path.join(td, "libaerolc_emul.so")
    cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "host_emul", "aerolc_emul.cpp"),
           "-L" + os.path.join(ROOT, "oracle"), "-l:liboracle.so", "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    subprocess.check_call(cmd)
    L = C.CDLL(so)
    L.emul_create.restype = C.c_void_p
    L.emul_create.argtypes = [C.c_int, C.c_int]
    L.emul_destroy.argtypes = [C.c_void_p]
    L.emul_write.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.emul_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.emul_tick.argtypes = [C.c_void_p]
    return L


def drain(L, h, ch, which, rowshape, dtype, cap=4096):
    buf = np.zeros((cap,) + rowshape, dtype=dtype)
    n = L.emul_read(h, ch, which, buf.ctypes.data, cap)
    return buf[:n]


def oracle_run(O, soft, group=32):
    a = O.AeroL(8400)
    for s in range(0, len(soft), group):
        a.write(soft[s:s + group])
    fn, voice = a.take_voice()
    return fn, voice, a.take_sus(), a.take_events()


@pytest.mark.parametrize("nch,write,splice", [(5, 3000, False), (70, 5000, False), (3, 9000, False), (9, 9000, True), (6, 16000, True)])
def test_bank_logic_vs_oracle(E, oracle_mod, nch, write, splice):
    """splice: copies of the unique word planted so that they END inside the detection window of a frame (pre-increment cntr > 3984) -- the
    reference then abandons the frame for a new one, and k_aerolc_bits meets a second or third jumpable stretch within one round (the
    third ends its round early): the bulk copies must land in stream order and the round count must still cover the write."""
    rng = np.random.default_rng(77 + nch)
    streams = []
    for c in range(nch):
        lead = int(rng.integers(0, 4200))
        frames, soft = AF.c_channel_case(5000 + c, 4 + c % 3, 10.0 + 5.0 * (c % 7), inv=(bool(c & 1), bool(c & 2)), lead=lead)
        if splice and c % 3 != 2:
            soft = soft.copy()
            uw = soft[lead:lead + 104].copy()
            for f in range(len(frames)):
                if (f + c) % 2 == 0:
                    # the detectors' shift registers move only inside the window (body bits 3986 ..), so the planted word must lie in it
                    # entirely, on the arm parity of the real one: it ends at body bit 4089 .. 4094
                    e = lead + f * 4200 + 104 + int(rng.integers(4089, 4094))
                    e += (e - 103 - lead) % 2
                    soft[e - 103:e + 1] = uw
                    if rng.random() < 0.7:  # and another one in the window of the displaced frame that this one starts
                        e2 = e + 1 + int(rng.integers(4089, 4094))
                        e2 += (e2 - 103 - lead) % 2
                        if e2 + 1 < len(soft):
                            soft[e2 - 103:e2 + 1] = uw
        streams.append(soft)
    h = E.emul_create(nch, 0)
    pos = [0] * nch
    while any(pos[c] < len(streams[c]) for c in range(nch)):
        cnt = np.array([min(int(rng.integers(write // 2, write + 1)), len(streams[c]) - pos[c]) for c in range(nch)], dtype=np.int32)
        buf = np.zeros((nch, write), dtype=np.int16)
        for c in range(nch):
            buf[c, :cnt[c]] = streams[c][pos[c]:pos[c] + cnt[c]]
            pos[c] += int(cnt[c])
        E.emul_write(h, buf.ctypes.data, cnt.ctypes.data, write, int(cnt.max()))
    extra = 0
    for c in range(nch):
        ofn, ovoice, osus, oev = oracle_run(oracle_mod, streams[c])
        extra += int((oev[:, 1] == 2).sum()) - len(ofn)
        v = drain(E, h, c, 1, (304,), np.uint8)
        fn = v[:, :4].copy().view(np.uint32).reshape(-1)
        assert np.array_equal(fn, ofn) and np.array_equal(v[:, 4:], ovoice), c
        assert np.array_equal(drain(E, h, c, 0, (16,), np.int32), osus), c
        assert np.array_equal(drain(E, h, c, 2, (3,), np.int64), oev), c
    E.emul_destroy(h)
    if splice:
        assert extra >= nch // 2, extra  # the planted words did fire: more unique words than completed frames
