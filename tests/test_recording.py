"""The reference's own sample recording of a 10.5 kbps P channel (samples/10.5k_sample.ogg) through the hot path.

tests/golden/recording_oqpsk_10k5.npz holds 12 s of COMMON INPUT made from it -- 48 kHz int16 PCM decoded by scripts/vorbis_decode.py (a Vorbis I
decoder written from the specification; it cannot be checked against libvorbis here, so this is "the PCM all three sides were fed", not "the
recording's PCM as the reference's audio stack would decode it") and resampled from 44.1 kHz: tests/golden/make_recording_golden.py -- and what
the UNMODIFIED reference made of exactly that PCM: the soft bits OqpskDemodulator handed over,
one status row per frequency estimate, and the signal units its AeroL printed (545, 468 of them CRC-clean; carrier found at 5757 Hz with the
default centre of 8000 Hz).  CPU: the restatement must reproduce all three exactly.  GPU: a bank fed the recording at several time offsets
against the oracle, and PCM -> demodulator bank -> Aero-L bank on the device must print the reference's CRC-clean signal units."""
import os

import numpy as np
import pytest

from conftest import assert_soft_bytes, load_golden

NAME = "recording_oqpsk_10k5"


def test_oracle_matches_reference_on_the_recording(oracle_mod):
    g = load_golden(NAME)
    o = oracle_mod.run_demod(oracle_mod.oqpsk_settings(), g["pcm"], chunk=4096)
    assert np.array_equal(o["soft"], g["soft"])
    assert o["status"].shape == g["status"].shape and np.array_equal(o["status"], g["status"])
    assert abs(g["status"][-1, 1] - 5757.3) < 1.0 and g["status"][-1, 5] == 1  # the carrier the reference settled on, signal present
    a = oracle_mod.run_aerol(10500, g["soft"])["sus"]
    assert a.shape[0] == g["sus"].shape[0] == 545
    assert np.array_equal(a[:, 1], g["sus"][:, 0]) and np.array_equal(a[:, 2:12], g["sus"][:, 1:11]) and np.array_equal(a[:, 14], g["sus"][:, 11])
    assert int(g["sus"][:, 11].sum()) == 468


def test_the_decoder_reproduces_the_fixture():
    """Where the reference tree and scipy are present: the first second of the fixture comes out of the Ogg file again (guards the decoder)."""
    path = "/root/reference/samples/10.5k_sample.ogg"
    if not os.path.exists(path):
        pytest.skip("reference samples not present on this machine")
    scipy_signal = pytest.importorskip("scipy.signal")
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import vorbis_decode

    x, rate = vorbis_decode.decode(path, 1.5)
    assert rate == 44100 and x.shape[0] == 1
    y = scipy_signal.resample_poly(x[0], 160, 147)
    pcm = np.clip(np.round(y * 32767.0), -32768, 32767).astype(np.int16)
    g = load_golden(NAME)
    assert np.array_equal(pcm[:48000], g["pcm"][:48000])


@pytest.mark.gpu
def test_gpu_bank_on_the_recording(oracle_mod):
    """Five channels carry the recording from different starting points (so symbol clocks, coarse-estimate instants and write boundaries fall
    differently in each): soft bits, soft symbols and every status row against the oracle; channel 0 (the fixture as it is) also against the
    reference's own soft bits; then the soft bits stay on the device and the Aero-L bank must print the reference's signal units."""
    from jaero_amd import capi
    from jaero_amd import demodulator as B

    capi.lib()
    g = load_golden(NAME)
    shifts = [0, 1234, 7777, 20001, 48000]
    n = len(g["pcm"]) - max(shifts)
    pcm = np.stack([g["pcm"][s:s + n] for s in shifts])
    nch, chunk = len(shifts), 4096
    demod = B.DemodulatorBank(B.OqpskSettings(), nch, device=0, ebno=True, status_log=True, capture_symbols=True, max_write_samples=chunk,
                              softbit_capacity=2 * n * 10500 // 48000 + 1024)
    for s in range(0, n, chunk):
        demod.write(pcm[:, s:s + chunk])
    worst = []
    for c in range(nch):
        ref = oracle_mod.run_demod(oracle_mod.oqpsk_settings(), pcm[c], chunk=chunk, capture_symbols=True)
        soft, sym, log = demod.read_softbits(c), demod.read_symbols(c), demod.read_status_log(c)
        m = len(ref["soft"])
        assert len(soft) == m + ref["pending"], c
        assert np.array_equal(soft[:m] >= 128, ref["soft"] >= 128), f"channel {c}: hard decisions differ"
        assert_soft_bytes(soft[:m], ref["soft"], f"channel {c}")
        assert sym.shape == ref["symbols"].shape, c
        d = np.abs(sym - ref["symbols"]).max(axis=1)
        worst.append(float(d.max(initial=0.0)))
        # The north star's tolerance on EVERY symbol of EVERY channel.  Until round 5 the channel starting 1234 samples in stepped 1.6e-4 away at
        # symbol 25 282 and drifted back: the device library's hypot differs from glibc's by an ulp in 14.5 % of the calls, and at that place of
        # the recording the loops amplify it (reproduced on the CPU by perturbing the oracle's hypot: DESIGN 9 item 18).  The kernels now call
        # glibc 2.35's hypot restated operation for operation and a correctly rounded atan2 (jaero_amd/csrc/jd_libm.h).
        assert d.max(initial=0.0) < 1e-5, (c, float(d.max(initial=0.0)), int(d.argmax()))
        assert log.shape == ref["status"].shape and np.array_equal(log[:, [0, 5]], ref["status"][:, [0, 5]]), c
        assert np.max(np.abs(log[:, 1:5] - ref["status"][:, 1:5])) < 1e-6, c
        if c == 0:
            k = min(m, len(g["soft"]))
            assert np.array_equal(soft[:k] >= 128, g["soft"][:k] >= 128)  # the unmodified reference's own decisions
    assert max(worst) < 1e-5, worst  # the north star's tolerance, on all five
    demod.close()
    # PCM -> soft bits -> signal units without leaving the device, half a second per write
    demod = B.DemodulatorBank(B.OqpskSettings(), nch, device=0, max_write_samples=24000, softbit_capacity=8192)
    aerol = B.AeroLBank(nch, 10500, max_softbits_per_write=8192, su_capacity=700)
    for s in range(0, n, 24000):
        demod.write(pcm[:, s:s + 24000])
        aerol.write_from_bank(demod, 8192)
    good_ref = [bytes(r[1:11].astype(np.uint8)) for r in g["sus"] if r[11]]
    counts = []
    for c in range(nch):
        sus = aerol.read_sus(c)
        good = [bytes(r[2:12].astype(np.uint8)) for r in sus if r[14]]
        # the oracle's chain on the same stream, written the same way
        d, al = oracle_mod.Demod(oracle_mod.oqpsk_settings()), oracle_mod.AeroL(10500)
        for s in range(0, n, 24000):
            d.write(pcm[c, s:s + 24000])
            al.write(d.take_soft())
        osus = al.take_sus()
        assert sus.shape == osus.shape and np.array_equal(sus[:, 1:15], osus[:, 1:15]), c  # every printed unit, clean or not
        counts.append(len(good))
        if c == 0:
            assert good == good_ref[:len(good)] and len(good) >= len(good_ref) - 54  # the reference's units in its order (these streams end a second = two frames earlier)
    # (where the stream starts decides how the reference acquires: from 1234 samples in it prints 519 units of which only 52 pass the CRC --
    # and so does this; the other starting points give 416 to 468)
    assert sorted(counts)[1] >= 400 and min(counts) >= 40, counts
    demod.close()
    aerol.close()
