"""The reference's own off-air recording of a 10.5 kbps R/T (burst) channel (samples/10.5k_burst_sample.mp3) through the burst hot path
(BASELINE configs[3]'s kind; VERDICT r5 missing #2: until round 6 burst OQPSK had only ever been fed synthetic bursts).

tests/golden/recording_burst_oqpsk_10k5.npz holds 15 s of COMMON INPUT made from it -- 48 kHz int16 PCM decoded by scripts/mp3_decode.py (an
MPEG-1 Layer III decoder written from the standard; this image has no audio decoder) and resampled from 44.1 kHz:
tests/golden/make_burst_recording_golden.py -- and what the UNMODIFIED reference made of exactly that PCM: the soft bits and start-of-burst markers
BurstOqpskDemodulator handed over (6 bursts in these 15 s, carrier found at ~11.1 kHz with the default centre of 8 kHz), its SignalStatus / EbNo /
Plottables emissions, and the packets its AeroL printed in burst mode: four T packets (10, 16, 17, 16 signal units, every one CRC-clean -- which
also says the decoder is right) and one " Bad R/T Packet".
CPU: the restatement must reproduce all of it exactly.  GPU: a bank fed the recording at several time offsets against the oracle (markers at the
same indices, hard bits equal, soft bytes counted, soft symbols within 1e-5, every emission at the same sample), then PCM -> burst demodulator bank
-> burst Aero-L bank on the device must print the reference's packets."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import assert_soft_bytes, load_golden

NAME = "recording_burst_oqpsk_10k5"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def burst_rows(packets):
    """[(type, bytes)] of the oracle / GPU -> the golden's row form (tests/test_aerol_oracle.py::burst_rows at 10500 bps)"""
    rows = []
    for typ, data in packets:
        if typ == 1:
            rows.append([1, 17, 0] + list(data[:17]) + [0] * (10 * 31 + 4 - 17))
        else:
            n = (len(data) + 1 - 6) // 12
            flat = [v for k in range(n) for v in data[6 + 12 * k: 6 + 12 * k + 10]]
            rows.append([2, n, n] + list(data[:4]) + flat + [0] * (10 * 31 - len(flat)))
    return np.array(rows, dtype=np.int32).reshape(-1, 317)


def test_fixture_is_what_the_reference_made_of_it():
    g = load_golden(NAME)
    info = json.loads(str(g["decoder"]))
    # the decoder's self-checks: never out of sync, the bit reservoir closes on every frame, every granule's Huffman data ends on its last bit
    assert info["resyncs"] == 0 and info["reservoir_underruns"] == 0 and info["reservoir_overlaps"] == 0 and info["huffman_misfits"] == 0 and info["granule_overruns"] == 0
    assert len(g["pcm"]) == 15 * 48000 and int((g["soft"] == -1).sum()) == 6
    assert [(int(r[0]), int(r[1])) for r in g["packets"]] == [(2, 10), (2, 16), (2, 17), (2, 16)] and int(g["bad"]) == 1
    assert bytes(g["packets"][0, 3:7].astype(np.uint8)).hex().upper() == "394A0E43"  # T packet from AES 394A0E to GES 43, as the reference prints it


def test_oracle_matches_reference_on_the_burst_recording(oracle_mod):
    g = load_golden(NAME)
    o = oracle_mod.run_burst(oracle_mod.burst_oqpsk_settings(), g["pcm"], chunk=4096)
    assert np.array_equal(o["soft"], g["soft"])
    ev = o["events"].copy()
    ev[:, 0] = np.floor(ev[:, 0] / 4096) * 4096  # the reference driver stamps an emission with the first sample of the write that carried it
    key = lambda e: e[np.lexsort((e[:, 1], e[:, 0]))]
    assert ev.shape == g["events"].shape and np.array_equal(key(ev)[:, :2], key(g["events"])[:, :2])
    assert np.max(np.abs(key(ev)[:, 2] - key(g["events"])[:, 2]) / np.maximum(1.0, np.abs(key(g["events"])[:, 2]))) < 1e-9
    a = oracle_mod.run_aerol_burst(10500, g["soft"])
    assert np.array_equal(burst_rows(oracle_mod.packets_from_rows(a["packets"])), g["packets"])
    assert int((a["events"][:, 1] == 3).sum()) == int(g["bad"])


def test_the_mp3_decoder_reproduces_the_fixture():
    """Where the reference tree and scipy are present: the first two seconds of the fixture come out of the MP3 file again (guards the decoder)."""
    path = "/root/reference/samples/10.5k_burst_sample.mp3"
    if not os.path.exists(path):
        pytest.skip("reference samples not present on this machine")
    scipy_signal = pytest.importorskip("scipy.signal")
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import mp3_decode
    import mp3_tables

    assert mp3_tables.check()
    x, rate, info = mp3_decode.decode(path, 2.6)
    assert rate == 44100 and x.shape[0] == 1 and info["huffman_misfits"] == 0 and info["reservoir_overlaps"] == 0 and info["resyncs"] == 0
    y = scipy_signal.resample_poly(x[0], 160, 147)
    pcm = np.clip(np.round(y * 32767.0), -32768, 32767).astype(np.int16)
    g = load_golden(NAME)
    assert np.array_equal(pcm[:96000], g["pcm"][:96000])


def test_mp3_decoder_rejects_a_damaged_table():
    """The self-check the decoder's correctness rests on: with ONE code length of ONE Huffman table changed the recording no longer decodes cleanly."""
    path = "/root/reference/samples/10.5k_burst_sample.mp3"
    if not os.path.exists(path):
        pytest.skip("reference samples not present on this machine")
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import importlib

    import mp3_decode
    import mp3_tables

    xs, lens, codes = mp3_tables.HUFF[15]
    j = 17
    saved = (lens[j], codes[j])
    try:
        # swap two code words of different lengths: the table stays a complete prefix code, only the assignment is wrong
        k = next(i for i in range(len(lens)) if lens[i] != lens[j])
        lens[j], codes[j], lens[k], codes[k] = lens[k], codes[k], lens[j], codes[j]
        importlib.reload(mp3_decode)
        try:
            _, _, info = mp3_decode.decode(path, 3.0)
            clean = info["huffman_misfits"] == 0
        except (ValueError, IndexError):
            clean = False
        assert not clean
    finally:
        k = next(i for i in range(len(lens)) if (lens[i], codes[i]) == saved)
        lens[j], codes[j], lens[k], codes[k] = lens[k], codes[k], lens[j], codes[j]
        importlib.reload(mp3_decode)


@pytest.mark.gpu
def test_gpu_burst_bank_on_the_recording(oracle_mod):
    """Four channels carry the recording from different starting points (so burst starts, trident windows, symbol instants and write boundaries fall
    differently in each), ragged write sizes: soft bits incl. markers, soft symbols and every emission against the oracle; channel 0 (the fixture as
    it is) also against the reference's own soft bits.  Then the soft bits stay on the device: burst demodulator bank -> burst Aero-L bank must print
    the reference's packets."""
    from jaero_amd import capi
    from jaero_amd import demodulator as B

    capi.lib()
    g = load_golden(NAME)
    shifts = [0, 1234, 7777, 20001]
    n = len(g["pcm"]) - max(shifts)
    pcm = np.stack([g["pcm"][s:s + n] for s in shifts])
    nch = len(shifts)
    bank = B.DemodulatorBank(B.BurstOqpskSettings(), nch, device=0, capture_symbols=True, trace=True, max_write_samples=5000, softbit_capacity=60000)
    rng = np.random.default_rng(6)
    s = 0
    while s < n:
        m = min(int(rng.integers(1, 5000)), n - s)
        bank.write(pcm[:, s:s + m])
        s += m
    nbursts = []
    differing = 0
    for c in range(nch):
        ref = oracle_mod.run_burst(oracle_mod.burst_oqpsk_settings(), pcm[c], chunk=4096, capture_symbols=True, trace=True)
        got = bank.read_softbits(c)
        assert len(got) == len(ref["soft"]), c
        assert np.array_equal(got == -1, ref["soft"] == -1), f"channel {c}: burst markers differ"
        assert np.array_equal(got >= 128, ref["soft"] >= 128), f"channel {c}: hard decisions differ"
        differing += int((got.astype(int) != ref["soft"].astype(int)).sum())
        assert_soft_bytes(got, ref["soft"], f"burst recording channel {c}", allow=2)
        ev, rev = bank.read_events(c), ref["events"]
        key = lambda e: e[np.lexsort((e[:, 1], e[:, 0]))]
        assert ev.shape == rev.shape and np.array_equal(key(ev)[:, :2], key(rev)[:, :2]), c
        assert np.max(np.abs(key(ev)[:, 2] - key(rev)[:, 2]) / np.maximum(1.0, np.abs(key(rev)[:, 2]))) < 1e-6, c
        sym = bank.read_symbols(c)
        assert sym.shape == ref["symbols"].shape, c
        assert np.max(np.abs(sym - ref["symbols"]), initial=0.0) < 1e-5, (c, float(np.max(np.abs(sym - ref["symbols"]))))
        nbursts.append(int((ref["soft"] == -1).sum()))
        if c == 0:
            assert np.array_equal(got >= 128, g["soft"] >= 128) and np.array_equal(got == -1, g["soft"] == -1)  # the unmodified reference's own stream
    assert nbursts[0] == 6 and min(nbursts) >= 5, nbursts
    bank.close()
    # PCM -> soft bits -> R/T packets without leaving the device
    demod = B.DemodulatorBank(B.BurstOqpskSettings(), nch, device=0, max_write_samples=4096, softbit_capacity=16384)
    aerol = B.AeroLBank(nch, 10500, max_softbits_per_write=16384, su_capacity=400, burst=True)
    for s in range(0, n, 4096):
        demod.write(pcm[:, s:s + 4096])
        aerol.write_from_bank(demod, 4096)
    for c in range(nch):
        soft = oracle_mod.run_burst(oracle_mod.burst_oqpsk_settings(), pcm[c], chunk=4096)["soft"]
        want = oracle_mod.packets_from_rows(oracle_mod.run_aerol_burst(10500, soft)["packets"])
        got = aerol.read_packets(c)
        assert got == want, c
        if c == 0:
            assert np.array_equal(burst_rows(got), g["packets"])  # what the reference's AeroL printed for this recording
    demod.close()
    aerol.close()
