"""GPU (-m gpu): the Aero-L bit pipeline (unique word / ambiguity -> deinterleave -> Viterbi -> delay line -> descramble -> CRC),
called through the C ABI, against the unmodified AeroL's goldens and the oracle; and the whole chain PCM -> demodulator bank ->
(device-resident soft bits) -> Aero-L bank -> CRC-clean signal units equal to the transmitted payloads.  Integer work: exact."""
import numpy as np
import pytest

from conftest import load_golden
from jaero_amd import aerol_frames as AF
from jaero_amd import signalgen as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    from jaero_amd import capi
    from jaero_amd import demodulator as D

    capi.lib()
    return D


def rows12(sus):
    return np.array([[int(r[1])] + [int(v) for v in r[2:12]] + [int(r[14])] for r in sus], dtype=np.int32).reshape(-1, 12)


@pytest.mark.parametrize("fb", [10500, 1200, 600])
def test_against_reference_golden(B, fb):
    g = load_golden(f"aerol_{fb}")
    soft, grp = g["soft"], int(g["group"])
    bank = B.AeroLBank(1, fb, max_softbits_per_write=8192)
    for s in range(0, len(soft), 4000):  # the unmodified AeroL was fed `grp` soft bits at a time; results are chunk invariant
        bank.write(soft[None, s:s + 4000])
    assert np.array_equal(rows12(bank.read_sus(0)), g["sus"])
    bank.close()


@pytest.mark.parametrize("fb,layout", [(10500, "wave"), (1200, "wave"), (10500, "lanes"), (600, "lanes")])
def test_bank_vs_oracle(B, oracle_mod, force_viterbi_layout, fb, layout):
    """70 channels (two wave groups), different frames / noise / arm inversions / garbage prefixes per channel, ragged per-channel
    counts in every write."""
    force_viterbi_layout(layout)  # large banks decode one block per lane; force it at this size too
    nch = 70
    rng = np.random.default_rng(fb)
    streams = []
    for c in range(nch):
        pay = AF.random_payloads(5, fb, seed=1000 + c)
        bits, _ = AF.p_channel_bits(pay, fb, invert_i=bool(c & 1), invert_q=bool(c & 2))
        pre = rng.integers(0, 2, size=int(rng.integers(0, 900)), dtype=np.uint8)
        streams.append(AF.to_soft(np.concatenate([pre, bits]), sigma=float(rng.uniform(0, 45)), seed=c))
    bank = B.AeroLBank(nch, fb, max_softbits_per_write=6000, su_capacity=400)
    pos = np.zeros(nch, dtype=np.int64)
    lens = np.array([len(s) for s in streams])
    while (pos < lens).any():
        cnt = np.minimum(rng.integers(1, 6000, size=nch), lens - pos).astype(np.int32)
        buf = np.zeros((nch, 6000), np.int16)
        for c in range(nch):
            buf[c, :cnt[c]] = streams[c][pos[c]:pos[c] + cnt[c]]
        bank.write(buf, cnt)
        pos += cnt
    nclean = 0
    for c in range(nch):
        o = oracle_mod.run_aerol(fb, streams[c], 1 << 20)
        assert np.array_equal(bank.read_sus(c), o["sus"]), c
        assert np.array_equal(bank.read_events(c), o["events"]), c
        nclean += int(o["sus"][:, 14].sum())
    assert nclean > nch * 10
    bank.close()


def test_unique_words_planted_in_the_detector_window(B, oracle_mod):
    """Round 6: a LOCKED 10.5 kbps channel's detector is on for the last 325 soft bits of a frame, and k_aerol_bits jumps that stretch too (after the 64 bits
    that flush the detector's shift registers) wherever k_aerol_scan saw no possible hit.  Here possible hits are planted there: the unique word or its complement
    on ONE arm only (a lone hit flips that arm's polarity flag for the bits that follow, aerol.cpp:781-804), on both arms one bit apart (a false sync inside the
    window), inside the first 64 bits of the window (where the registers still hold bits from before the frame body), and straddling the point where the
    jump may begin.  Every channel against the oracle fed the same stream in the same ragged writes; two write widths so that windows meet write boundaries."""
    fb, nch = 10500, 66
    rng = np.random.default_rng(66)
    uw = np.array([(0xE15AE893 >> (31 - k)) & 1 for k in range(32)], dtype=np.uint8)
    streams = []
    for c in range(nch):
        pay = AF.random_payloads(6, fb, seed=3000 + c)
        bits, _ = AF.p_channel_bits(pay, fb, invert_i=bool(c & 1), invert_q=bool(c & 2))
        bits = bits.copy()
        pre = rng.integers(0, 2, size=int(rng.integers(0, 700)), dtype=np.uint8)
        # frame f occupies bits[5250 f : 5250 (f + 1)]: 64 unique-word bits (32 per arm), then 16 + 178 + 4992; the detector window of frame f are its last
        # 325 - 64 = 261 body bits (pre-increment cntr 4925 .. 5185 counts from the end of the unique word)
        for f in range(1, 5):
            body_end = 5250 * (f + 1)                   # first bit of the NEXT frame's unique word
            win0 = body_end - 261
            kind = (c + f) % 6
            if kind == 0:
                continue
            start = {1: win0 + 70, 2: win0 + 71, 3: win0 + 10, 4: win0 + 40, 5: win0 + 150}[kind] + int(rng.integers(0, 8)) * 2
            word = uw if (c + f) & 1 else 1 - uw
            arm = bits[start:start + 64:2]
            if len(arm) == 32 and start + 64 < body_end - 2:
                bits[start:start + 64:2] = word                           # one arm
                if kind in (2, 5):
                    bits[start + 1:start + 65:2] = word if kind == 2 else 1 - word  # the other arm, one bit later: a false sync (either polarity)
        streams.append(AF.to_soft(np.concatenate([pre, bits]), sigma=float(rng.uniform(0, 20)), seed=c))
    for width in (6000, 1777):
        bank = B.AeroLBank(nch, fb, max_softbits_per_write=width, su_capacity=400)
        pos = np.zeros(nch, dtype=np.int64)
        lens = np.array([len(x) for x in streams])
        wr = np.random.default_rng(width)
        while (pos < lens).any():
            cnt = np.minimum(wr.integers(max(1, width // 2), width, size=nch), lens - pos).astype(np.int32)
            buf = np.zeros((nch, width), np.int16)
            for c in range(nch):
                buf[c, :cnt[c]] = streams[c][pos[c]:pos[c] + cnt[c]]
            bank.write(buf, cnt)
            pos += cnt
        nclean = nev = 0
        for c in range(nch):
            o = oracle_mod.run_aerol(fb, streams[c], 1 << 20)
            assert np.array_equal(bank.read_sus(c), o["sus"]), (width, c)
            assert np.array_equal(bank.read_events(c), o["events"]), (width, c)
            nclean += int(o["sus"][:, 14].sum())
            nev += len(o["events"])
        assert nclean > nch * 8 and nev > nch * 8  # frames without a planted word still decode; the planted hits show up as frame-length / sync events
        bank.close()


def test_start_of_burst_markers(B, oracle_mod):
    """Negative soft values (the burst demodulators' start-of-burst marker, aerol.cpp:1146-1152) in some channels' input: those
    channels are walked bit by bit for that write, their neighbours in the same wavefront keep jumping over locked frame bodies."""
    fb, nch = 10500, 12
    rng = np.random.default_rng(5)
    streams = []
    for c in range(nch):
        bits, _ = AF.p_channel_bits(AF.random_payloads(5, fb, seed=300 + c), fb, invert_i=bool(c & 1), invert_q=bool(c & 2))
        x = AF.to_soft(bits, sigma=20.0, seed=c)
        if c % 3 == 0:
            x = np.insert(x, np.sort(rng.integers(0, len(x), size=4)), -1)
        streams.append(x)
    n = max(len(x) for x in streams)
    bank = B.AeroLBank(nch, fb, max_softbits_per_write=4096, su_capacity=400)
    for s0 in range(0, n, 4096):
        buf = np.zeros((nch, 4096), np.int16)
        cnt = np.zeros(nch, np.int32)
        for c in range(nch):
            seg = streams[c][s0:s0 + 4096]
            buf[c, :len(seg)] = seg
            cnt[c] = len(seg)
        bank.write(buf, cnt)
    for c in range(nch):
        o = oracle_mod.run_aerol(fb, streams[c], 1 << 20)
        assert np.array_equal(bank.read_sus(c), o["sus"]), c
        assert np.array_equal(bank.read_events(c), o["events"]), c
    bank.close()


def test_carrier_loss_noise_and_reacquisition(B, oracle_mod):
    """Channels that lose their carrier (updateDCD ticks drop DataCarrierDetect), free-run over noise -- decoding junk blocks, as the
    reference does -- and re-acquire; and channels that never see anything but noise.  Unlocked channels jump over the stretches
    between the positions k_aerol_scan marks as possible unique words; one channel gets a planted unique word in its noise."""
    fb, nch, W = 10500, 9, 5250
    rng = np.random.default_rng(11)
    streams = []
    for c in range(nch):
        noise = lambda n: np.clip(np.round(128 + rng.normal(0, 45, n)), 0, 255).astype(np.int16)
        if c % 3 == 2:
            x = noise(11 * W)
            if c == 5:  # a perfect unique word on both arms in the middle of noise: a false sync, then junk frames
                uw = np.repeat([(AF.UW >> (31 - k)) & 1 for k in range(32)], 2)
                x[20011:20011 + 64] = np.where(uw > 0, 200, 55)
        else:
            a, _ = AF.p_channel_bits(AF.random_payloads(3, fb, seed=500 + c), fb, invert_i=bool(c & 1), invert_q=bool(c & 2))
            b, _ = AF.p_channel_bits(AF.random_payloads(3, fb, seed=600 + c), fb, first_counter=7)
            x = np.concatenate([noise(int(rng.integers(0, 700))), AF.to_soft(a, sigma=25.0, seed=c), noise(4 * W + int(rng.integers(0, 999))),
                                AF.to_soft(b, sigma=25.0, seed=c + 50)])
        streams.append(x)
    n = max(len(x) for x in streams)
    bank = B.AeroLBank(nch, fb, max_softbits_per_write=W, su_capacity=600)
    orc = [oracle_mod.AeroL(fb) for _ in range(nch)]
    step = 0
    for s0 in range(0, n, W):
        buf = np.zeros((nch, W), np.int16)
        cnt = np.zeros(nch, np.int32)
        for c in range(nch):
            seg = streams[c][s0:s0 + W]
            buf[c, :len(seg)] = seg
            cnt[c] = len(seg)
            for g0 in range(0, len(seg), 32):
                orc[c].write(seg[g0:g0 + 32])
        bank.write(buf, cnt)
        step += 1
        if step >= 4:  # from the fourth write on the 1 s timer fires twice per write: the carrier detect of silent channels drops
            for _ in range(2):
                d = bank.tick_dcd()
                for c in range(nch):
                    assert orc[c].tick_dcd() == int(d[c])
    nclean = 0
    for c in range(nch):
        sus, ev = orc[c].take_sus(), orc[c].take_events()
        assert np.array_equal(bank.read_sus(c, 600), sus), c
        assert np.array_equal(bank.read_events(c), ev), c
        nclean += int(sus[:, 14].sum()) if len(sus) else 0
    assert nclean > 100
    bank.close()


def test_tick_dcd(B):
    g = load_golden("aerol_1200")
    bank = B.AeroLBank(1, 1200, max_softbits_per_write=16384)
    bank.write(g["soft"][None, :])
    d = [int(bank.tick_dcd()[0]) for _ in range(6)]  # updateDCD: countdown 12 -> 0 in steps of 3, then DataCarrierDetect(false)
    assert d[0] == 1 and d[-1] == 0
    ev = bank.read_events(0)
    assert ev[-1, 1] == 0 and ev[-1, 2] == 0
    bank.close()


def test_pcm_to_signal_units_on_device(B):
    """10.5 kbps P-channel frames -> OQPSK passband PCM -> continuous demodulator bank -> soft bits stay in HBM -> Aero-L bank.
    The recovered, CRC-clean signal units must be exactly the transmitted ones."""
    fb, nch, nfr = 10500, 3, 14
    pays, pcms = [], []
    for c in range(nch):
        pay = AF.random_payloads(nfr, fb, seed=50 + c)
        bits, _ = AF.p_channel_bits(pay, fb)
        n = int(len(bits) / 2 * 48000 / 5250) + 2000
        pcm, _ = G.oqpsk(n, fc=8000.0 + 11.0 * c, ebno_db=13.0, seed=70 + c, bits=np.concatenate([bits, np.zeros(64, np.uint8)]))
        pays.append(pay)
        pcms.append(pcm)
    n = min(len(p) for p in pcms)
    pcm = np.stack([p[:n] for p in pcms])
    chunk = 24000  # half a second = one frame
    demod = B.DemodulatorBank(B.OqpskSettings(), nch, device=0, max_write_samples=chunk, softbit_capacity=8192)
    aerol = B.AeroLBank(nch, fb, max_softbits_per_write=8192, su_capacity=26 * nfr + 8)
    for s in range(0, n, chunk):
        demod.write(pcm[:, s:s + chunk])
        aerol.write_from_bank(demod, 8192)
    for c in range(nch):
        sus = aerol.read_sus(c)
        good = [bytes(r[2:12].astype(np.uint8)) for r in sus if r[14]]
        sent = [p for fr in pays[c] for p in fr]
        assert len(good) >= 26 * 6, (c, len(good))
        # in order, a contiguous run of the transmitted units
        i0 = sent.index(good[0])
        assert good == sent[i0:i0 + len(good)], c
    demod.close()
    aerol.close()
