"""CPU: the Aero-L bit-pipeline restatement (oracle/aerol_oracle.c) pinned against what the UNMODIFIED AeroL printed
(tests/golden/aerol_*.npz, made by oracle/_ref) and, where oracle/_ref can run, against AeroL itself on fresh frames."""
import os

import numpy as np
import pytest

from conftest import load_golden
from jaero_amd import aerol_frames as AF


def oracle_rows(sus):
    return np.array([[int(r[1])] + [int(v) for v in r[2:12]] + [int(r[14])] for r in sus], dtype=np.int32).reshape(-1, 12)


@pytest.mark.parametrize("fb", [10500, 1200, 600])
def test_oracle_matches_reference_golden(oracle_mod, fb):
    g = load_golden(f"aerol_{fb}")
    o = oracle_mod.run_aerol(fb, g["soft"], int(g["group"]))
    assert np.array_equal(oracle_rows(o["sus"]), g["sus"])
    # the fixture is not trivial: frames synchronise and most units come back clean
    assert int(g["sus"][:, 11].sum()) >= 20


@pytest.mark.parametrize("fb", [10500, 1200, 600])
def test_generator_round_trip(oracle_mod, fb):
    """Clean frames: every signal unit of every frame but the start-up ones comes back with its CRC intact and its payload."""
    pay = AF.random_payloads(8, fb, seed=11)
    bits, flen = AF.p_channel_bits(pay, fb)
    o = oracle_mod.run_aerol(fb, AF.to_soft(bits), 32)
    sus = o["sus"]
    good = {(int(r[0]), int(r[1])): bytes(r[2:12].astype(np.uint8)) for r in sus if r[14]}
    d = AF.geometry(fb)["delay_frames"]
    nsu = len(pay[0])
    hits = 0
    for f in range(1, 8 - d):  # output frame f+d carries transmitted frame f (frame 0 suffers the decoder's start-up)
        for k in range(nsu):
            assert good.get((f + d, k)) == pay[f][k], (fb, f, k)
            hits += 1
    assert hits >= nsu * 3
    ev = o["events"]
    assert (ev[:, 1] == 2).sum() == 8  # one unique word per frame
    assert o["dcd"] == 1


def test_chunking_invariance(oracle_mod):
    g = load_golden("aerol_1200")
    a = oracle_mod.run_aerol(1200, g["soft"], 12)
    b = oracle_mod.run_aerol(1200, g["soft"], 1000)
    assert np.array_equal(a["sus"], b["sus"]) and np.array_equal(a["events"], b["events"])


@pytest.fixture(scope="module")
def R(oracle_mod):
    if not oracle_mod.have_ref():
        pytest.skip("oracle/_ref/jaero_ref not available here")
    try:
        oracle_mod.run_ref_aerol(1200, np.zeros(24, np.int16), 12)
    except Exception as e:
        pytest.skip(f"_ref cannot run here: {e}")
    return oracle_mod


@pytest.mark.parametrize("fb,sigma,inv", [(10500, 35.0, (False, True)), (10500, 10.0, (True, True)), (1200, 30.0, (False, False)), (600, 40.0, (False, False))])
def test_oracle_vs_unmodified_aerol(R, fb, sigma, inv):
    pay = AF.random_payloads(6, fb, seed=int(sigma) + fb)
    bits, _ = AF.p_channel_bits(pay, fb, invert_i=inv[0], invert_q=inv[1])
    # garbage before the first frame and a cut in the middle (short frame) to exercise resynchronisation
    rng = np.random.default_rng(fb)
    pre = rng.integers(0, 2, size=777, dtype=np.uint8)
    cut = np.concatenate([pre, bits[: 3 * len(bits) // 6 + 333], bits[4 * len(bits) // 6:]])
    soft = AF.to_soft(cut, sigma=sigma, seed=fb + 5)
    grp = 32 if fb == 10500 else 12
    ref, _ = R.run_ref_aerol(fb, soft, grp)
    o = R.run_aerol(fb, soft, grp)
    mine = [(int(r[1]), bytes(r[2:12].astype(np.uint8)), bool(r[14])) for r in o["sus"]]
    assert ref == mine


# ---------------------------------------------------------------------------------------------- burst mode (R / T channel packets)
def burst_rows(packets, msk=False):
    """oracle / GPU packet list [(type, bytes)] -> the golden files' row form [type, n, n_printed, header/payload bytes ...].
    The reference prints `numberofsus` signal units of a T packet: all of them at 10500 bps, one less than the block holds at 600 / 1200
    bps (updateMSK sets numberofsus = targetSUSize, aerol.h:759)."""
    rows = []
    for typ, data in packets:
        if typ == 1:
            rows.append([1, 17, 0] + list(data[:17]) + [0] * (10 * 31 + 4 - 17))
        else:
            n = (len(data) + 1 - 6) // 12 - (1 if msk else 0)
            flat = [v for k in range(n) for v in data[6 + 12 * k: 6 + 12 * k + 10]]
            rows.append([2, n, n] + list(data[:4]) + flat + [0] * (10 * 31 - len(flat)))
    return np.array(rows, dtype=np.int32).reshape(-1, 3 + 4 + 310)


@pytest.mark.parametrize("name", ["10500_a", "10500_b", "1200_a", "600_a"])
def test_burst_oracle_matches_reference_golden(oracle_mod, name):
    """R/T packets: the oracle's burst mode against what the unmodified AeroL (setSettings(10500, true)) printed for the same soft
    bits in the same (burst demodulator) groups: packet bytes, ' Bad R/T Packet' notices, DataCarrierDetect edges."""
    g = load_golden(f"aerol_burst_{name}")
    fb = int(name.split("_")[0])
    o = oracle_mod.run_aerol_burst(fb, g["soft"])
    assert np.array_equal(burst_rows(oracle_mod.packets_from_rows(o["packets"]), msk=fb != 10500), g["packets"])
    ev = o["events"]
    assert int((ev[:, 1] == 3).sum()) == int(g["bad"])
    # the driver stamps a DCD edge with the first soft bit of the group that carried it
    starts = np.array([s for s, _ in oracle_mod.demod_groups(g["soft"])])
    dcd = [(int(v), int(starts[np.searchsorted(starts, i, side="right") - 1])) for i, k, v in ev[1:] if k == 0]
    assert dcd == [tuple(r) for r in g["dcd"].tolist()]


def test_burst_generator_round_trip(oracle_mod):
    rng = np.random.default_rng(9)
    rb = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))
    pk = [("T", (rb(4), [rb(10) for _ in range(5)])), ("R", rb(17))]
    x = AF.rt_burst_stream(pk, sigma=25.0, seed=4, invert_i=True, invert_q=True)
    got = oracle_mod.packets_from_rows(oracle_mod.run_aerol_burst(10500, x)["packets"])
    assert [t for t, _ in got] == [2, 1]
    assert got[1][1][:17] == pk[1][1] and got[0][1][:4] == pk[0][1][0]
    assert [got[0][1][6 + 12 * k: 16 + 12 * k] for k in range(5)] == pk[0][1][1]


def test_burst_oracle_vs_unmodified_aerol(R):
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    for seed, sigma, inv, cut in ((5, 35.0, (False, True), True), (6, 12.0, (True, True), False)):
        _, x = mk.rt_case(seed, sigma, inv, cut)
        ref, bad, _ = R.run_ref_aerol_burst(10500, x)
        o = R.run_aerol_burst(10500, x)
        got = burst_rows(R.packets_from_rows(o["packets"]))
        want = []
        for p in ref:
            if p[0] == "R":
                want.append([1, 17, 0] + list(p[1]) + [0] * (10 * 31 + 4 - 17))
            else:
                flat = [v for su in p[3] for v in su]
                want.append([2, len(p[3]), p[2]] + list(p[1]) + flat + [0] * (10 * 31 - len(flat)))
        assert np.array_equal(got, np.array(want, dtype=np.int32).reshape(-1, 317))
        assert int((o["events"][:, 1] == 3).sum()) == bad
    for fb, seed in ((1200, 7), (600, 8)):
        _, x = mk.rt_case_msk(seed, 30.0, invert=bool(seed & 1), cut=True)
        ref, bad, _ = R.run_ref_aerol_burst(fb, x)
        o = R.run_aerol_burst(fb, x)
        got = burst_rows(R.packets_from_rows(o["packets"]), msk=True)
        assert np.array_equal(got, mk.ref_rows(ref))
        assert int((o["events"][:, 1] == 3).sum()) == bad


# ------------------------------------------------------------------------------------------------ 8400 bps C channel (DecodeC)
def c_run(O, soft, group=32):
    a = O.AeroL(8400)
    for s in range(0, len(soft), group):
        a.write(soft[s:s + group])
    fn, voice = a.take_voice()
    sus = a.take_sus()
    return fn, voice, sus, c_printed(sus), a


def c_printed(sus):
    """What DecodeC prints of the oracle's sub-band rows [frame, k, 12 bytes, crc_ok, 0]: the units with a good CRC that are not fill-in units
    -- of the frames at whose end the data carrier is detected: AeroL::DecodeC clears the text of a call that ends with datacd false
    (aerol.cpp:2497-2500), and datacd follows the CRC verdicts (+2 up to 12 for a good one, -5 for a bad one, high above 2: aerol.cpp:2370-2384;
    the 1 s timer that lowers it again never fires in the reference driver).  A frame's three units are checked within one call."""
    cd, dcd, out, k = 0, False, [], 0
    while k < len(sus):
        frame, fr = int(sus[k][0]), []
        while k < len(sus) and int(sus[k][0]) == frame:
            r = sus[k]
            if r[14]:
                if cd < 12:
                    cd += 2
            elif cd > 0:
                cd -= 5
            if not dcd and cd > 2:
                dcd = True
            if r[14] and r[2] != 0x01:
                fr.append(bytes(r[2:12].astype(np.uint8)))
            k += 1
        if dcd:
            out += fr
    return out


def c_voice_equal(ref_voice, voice):
    """Row 1 (the first frame after start-up) is compared without the three low bits of its last byte: the reference reads bits of
    JConvolutionalCodec::decoded that libcorrect did not write on the codec's first call (no overlap yet)."""
    assert ref_voice.shape == voice.shape
    a, b = ref_voice.copy(), voice.copy()
    if len(a) > 1:
        a[1, 299] &= 0xF8
        b[1, 299] &= 0xF8
    return np.array_equal(a, b)


def test_c_channel_oracle_matches_reference_golden(oracle_mod):
    g = load_golden("aerol_c_8400_a")
    fn, voice, sus, printed, _ = c_run(oracle_mod, g["soft"], int(g["group"]))
    assert c_voice_equal(g["voice"], voice)
    assert printed == [bytes(r) for r in g["sus"]]
    assert len(printed) >= 8  # the fixture synchronises and carries signal units


@pytest.mark.parametrize("inv", [(False, False), (True, False), (False, True), (True, True)])
def test_c_channel_generator_round_trip(oracle_mod, inv):
    """Frame f comes out while frame f + 1 is received; the codec's first call has no overlap, so frame 0 is lost (31 bits out of
    step) and every later one comes back exactly: 300 voice bytes and three signal units with their CRC."""
    frames, soft = AF.c_channel_case(900 + 2 * inv[0] + inv[1], 6, 18.0, inv=inv, lead=37)
    fn, voice, sus, printed, a = c_run(oracle_mod, soft)
    assert len(voice) == 6 and list(fn) == list(range(6))
    for k in range(2, 6):
        assert np.array_equal(voice[k], frames[k - 1][0])
        rows = sus[sus[:, 0] == k]
        assert [bytes(r[2:12].astype(np.uint8)) for r in rows] == frames[k - 1][1] and all(rows[:, 14] == 1)
    assert a.dcd == 1
    # chunking of the soft-bit stream does not matter
    fn2, voice2, sus2, _, _ = c_run(oracle_mod, soft, group=777)
    assert np.array_equal(voice, voice2) and np.array_equal(sus, sus2)


@pytest.mark.parametrize("seed,sigma,inv,lead", [(41, 10.0, (False, False), 74), (42, 30.0, (True, True), 11), (43, 45.0, (False, True), 200)])
def test_c_channel_oracle_vs_unmodified_aerol(R, seed, sigma, inv, lead):
    frames, soft = AF.c_channel_case(seed, 6, sigma, inv=inv, lead=lead)
    voice_ref, sus_ref, dcd_ref, _ = R.run_ref_aerol_c(soft, 32)
    fn, voice, sus, printed, _ = c_run(R, soft)
    assert c_voice_equal(voice_ref, voice)
    assert printed == sus_ref


@pytest.mark.parametrize("seed,sigma,inv,lead", [(51, 10.0, (False, False), 74), (52, 20.0, (True, False), 1501), (53, 25.0, (True, True), 11)])
def test_c_channel_false_unique_words_vs_unmodified_aerol(R, seed, sigma, inv, lead):
    """Copies of the unique word planted so that they lie entirely inside the detection window of a frame (body bits 3986 .. 4094, on the arm parity of
    the real one): the unmodified AeroL::DecodeC abandons the frame for a new one (aerol.cpp:2201-2316).  The restatement must print the same voice
    frames and signal units -- it is what tests/test_aerolc_emul.py and the GPU bank tests compare k_aerolc_bits / k_aerolc_bulk with in exactly
    this situation (a second and third jumped stretch within one round)."""
    rng = np.random.default_rng(seed)
    frames, soft = AF.c_channel_case(seed, 7, sigma, inv=inv, lead=lead)
    soft = soft.copy()
    uw = soft[lead:lead + 104].copy()
    planted = 0
    for f in range(len(frames)):
        if f % 2 == 0:
            e = lead + f * 4200 + 104 + int(rng.integers(4089, 4094))
            e += (e - 103 - lead) % 2
            soft[e - 103:e + 1] = uw
            planted += 1
    voice_ref, sus_ref, dcd_ref, _ = R.run_ref_aerol_c(soft, 32)
    fn, voice, sus, printed, _ = c_run(R, soft)
    assert c_voice_equal(voice_ref, voice)
    assert printed == sus_ref
    a = R.AeroL(8400)
    a.write(soft)
    ev = a.take_events()
    assert int((ev[:, 1] == 2).sum()) > len(fn), (planted, len(fn))  # more unique words than completed frames: the planted ones fired

