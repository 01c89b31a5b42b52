"""GPU (-m gpu): the Aero-L C-channel bit pipeline (AeroL::DecodeC) through the C ABI against the oracle and the
reference golden: a single channel, banks of 5 and 70 channels with ragged writes, and the overflow report."""

import numpy as np
import pytest

from conftest import load_golden
from jaero_amd import aerol_frames as AF

pytestmark = pytest.mark.gpu


@pytest.fixture()
def D():
    from jaero_amd import capi
    from jaero_amd import demodulator as Dm

    capi.lib()
    return Dm


def oracle_run(O, soft, group=32):
    a = O.AeroL(8400)
    for s in range(0, len(soft), group):
        a.write(soft[s:s + group])
    fn, voice = a.take_voice()
    return fn, voice, a.take_sus(), a.take_events(), a


def test_golden_single_channel(D, oracle_mod):
    g = load_golden("aerol_c_8400_a")
    soft = g["soft"]
    bank = D.AeroLBank(1, 8400, max_softbits_per_write=4096)
    for s in range(0, len(soft), 4096):
        bank.write(soft[s:s + 4096].reshape(1, -1))
    fn, voice = bank.read_voice(0)
    sus = bank.read_sus(0)
    ofn, ovoice, osus, oev, _ = oracle_run(oracle_mod, soft)
    assert np.array_equal(fn, ofn) and np.array_equal(voice, ovoice)
    assert np.array_equal(sus, osus)
    ev = bank.read_events(0)
    assert np.array_equal(ev, oev)
    # and the reference itself (all but the three bits of row 1 it reads from memory libcorrect never wrote)
    a, b = g["voice"].copy(), voice.copy()
    a[1, 299] &= 0xF8
    b[1, 299] &= 0xF8
    assert np.array_equal(a, b)
    printed = [bytes(r[2:12].astype(np.uint8)) for r in sus if r[14] and r[2] != 0x01]
    assert printed == [bytes(r) for r in g["sus"]]
    bank.close()


@pytest.mark.parametrize("nch,write,layout", [(5, 3000, "wave"), (70, 5000, "wave"), (70, 5000, "lanes")])
def test_bank_vs_oracle(D, oracle_mod, force_viterbi_layout, nch, write, layout):
    """Channels at different frame phases, inversions and noise levels, ragged writes: every channel equals its own oracle run.  Both
    Viterbi layouts (one block per wavefront; one per lane, what banks of 16 384 channels and more use)."""
    force_viterbi_layout(layout)
    rng = np.random.default_rng(77 + nch)
    streams = []
    for c in range(nch):
        frames, soft = AF.c_channel_case(5000 + c, 4 + c % 3, 10.0 + 5.0 * (c % 7), inv=(bool(c & 1), bool(c & 2)), lead=int(rng.integers(0, 4200)))
        streams.append(soft)
    n = max(len(s) for s in streams)
    bank = D.AeroLBank(nch, 8400, max_softbits_per_write=write)
    pos = [0] * nch
    while any(pos[c] < len(streams[c]) for c in range(nch)):
        cnt = np.array([min(int(rng.integers(write // 2, write + 1)), len(streams[c]) - pos[c]) for c in range(nch)], dtype=np.int32)
        buf = np.zeros((nch, write), dtype=np.int16)
        for c in range(nch):
            buf[c, :cnt[c]] = streams[c][pos[c]:pos[c] + cnt[c]]
            pos[c] += int(cnt[c])
        bank.write(buf, counts=cnt)
    check = range(nch) if nch <= 8 else sorted({0, 1, 31, 63, 64, nch - 1})
    for c in check:
        ofn, ovoice, osus, oev, oa = oracle_run(oracle_mod, streams[c])
        fn, voice = bank.read_voice(c)
        assert np.array_equal(fn, ofn) and np.array_equal(voice, ovoice), c
        assert np.array_equal(bank.read_sus(c), osus), c
        assert np.array_equal(bank.read_events(c), oev), c
    bank.close()


def test_overflow_is_reported(D):
    """A caller that falls behind: with room for 3 signal units (= 1 voice frame) per channel the second frame's rows are dropped, and
    the next read says so once (JAERO_EOVERFLOW), as the P and R/T banks do."""
    from jaero_amd import capi

    frames, soft = AF.c_channel_case(6001, 4, 30.0, inv=(False, False), lead=100)
    bank = D.AeroLBank(1, 8400, max_softbits_per_write=4096, su_capacity=3)
    for s in range(0, len(soft), 4096):
        bank.write(soft[s:s + 4096].reshape(1, -1))
    with pytest.raises(capi.JaeroError) as e:
        bank.read_sus(0)
    assert e.value.code == capi.E_OVERFLOW
    assert len(bank.read_sus(0)) == 0  # reported once; the three rows of the first frame were handed over by the failing call's copy
    with pytest.raises(capi.JaeroError):
        bank.read_voice(0)
    bank.close()
