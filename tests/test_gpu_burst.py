"""GPU (-m gpu): the burst HIP path (Hilbert FIR -> front end -> trident FFT check -> tracking chain), called through the C ABI,
against the committed reference goldens and against the oracle on seeded multi-channel banks.

Contract (BASELINE.json north_star): hard decisions of the soft-bit stream bit-exact (incl. the -1 start-of-burst markers and
their positions), soft symbols within 1e-5; every SignalStatus / EbNo / Plottables emission at the same sample."""
import numpy as np
import pytest

from conftest import assert_soft_bytes, load_golden
from jaero_amd import signalgen as G

pytestmark = pytest.mark.gpu
SYM_TOL = 1e-5


@pytest.fixture(scope="module")
def B():
    from jaero_amd import capi
    from jaero_amd import demodulator as D

    capi.lib()
    return D


def bank_for(B, kind, opts, nch, **kw):
    if kind == "burstoqpsk":
        s = B.BurstOqpskSettings(freq_center=opts.get("freq_center", 8000.0))
    else:
        s = B.BurstMskSettings(freq_center=opts.get("freq_center", 1000.0), fb=float(opts.get("fb", 1200)),
                               lockingbw=opts.get("lockingbw", 1800.0))
    return B.DemodulatorBank(s, nch, device=0, **kw)


def oracle_settings(O, kind, opts):
    if kind == "burstoqpsk":
        return O.burst_oqpsk_settings(freq_center=opts.get("freq_center", 8000.0))
    return O.burst_msk_settings(freq_center=opts.get("freq_center", 1000.0), fb=float(opts.get("fb", 1200)),
                                lockingbw=opts.get("lockingbw", 1800.0))


def sort_ev(ev):
    return ev[np.lexsort((ev[:, 1], ev[:, 0]))] if len(ev) else ev


def check_soft(got, ref, tag="", allow=None):
    """The burst demodulators' input passes an FFT filter (the Hilbert transform) whose round-off is not the reference FFT's, so a soft byte on
    a rounding edge may differ by one: bounded by 1, counted in the session's ledger, and at most BURST_SOFT_ALLOW bytes of a stream."""
    assert len(got) == len(ref), tag
    assert np.array_equal(got == -1, ref == -1), "burst markers differ"
    assert np.array_equal(got >= 128, ref >= 128), "hard decisions differ"
    assert_soft_bytes(got, ref, tag, allow=(BURST_SOFT_ALLOW if allow is None else allow))


BURST_SOFT_ALLOW = 2  # per stream; the suite sees 1 differing byte among 1.1 million burst soft bytes (profiles/r5_soft_byte_ledger.json)


def check_events(got, ref):
    got, ref = sort_ev(got), sort_ev(ref)
    assert got.shape == ref.shape
    assert np.array_equal(got[:, :2], ref[:, :2]), "emission kinds / sample stamps differ"
    if len(ref):
        assert np.max(np.abs(got[:, 2] - ref[:, 2]) / np.maximum(1.0, np.abs(ref[:, 2]))) < 1e-6


@pytest.mark.parametrize("name", ["burst_oqpsk_10k5_default", "burst_oqpsk_10k5_chunk1500", "burst_msk_1200_sample1_excerpt"])
def test_against_reference_golden(B, name):
    """Same inputs the unmodified reference was run on (tests/golden, made by oracle/_ref), same write sizes."""
    g = load_golden(name)
    chunk = g["opts"].get("chunk", 4096)
    bank = bank_for(B, g["kind"], g["opts"], 1, max_write_samples=chunk, softbit_capacity=40000)
    pcm = g["pcm"]
    for s in range(0, len(pcm), chunk):
        bank.write(pcm[None, s:s + chunk])
    check_soft(bank.read_softbits(0), g["soft"])
    ev = bank.read_events(0)
    ev[:, 0] = np.floor(ev[:, 0] / chunk) * chunk  # the reference driver stamps emissions with their write's first sample
    check_events(ev, g["events"])
    st = bank.read_status(0)
    assert st.signal in (0, 1) and st.freq_est > 0
    bank.close()


def feed_random_chunks(bank, pcm, rng, lo, hi, layout):
    from jaero_amd import capi

    s, n = 0, pcm.shape[1]
    while s < n:
        m = min(int(rng.integers(lo, hi)), n - s)
        blk = pcm[:, s:s + m]
        if layout == capi.PCM_FRAME_MAJOR:
            bank.write(np.ascontiguousarray(blk.T), layout=layout)
        else:
            bank.write(blk, layout=layout)
        s += m


@pytest.mark.parametrize("layout", [0, 1])
def test_burst_oqpsk_bank_vs_oracle(B, oracle_mod, layout):
    """70 channels (two wave-groups, the second padded), per-channel carriers and burst times, ragged write sizes (so trident
    events, burst starts and symbol instants fall anywhere relative to the segment boundaries), both PCM layouts."""
    nch, n = 70, 120000
    rng = np.random.default_rng(1234 + layout)
    pcm = np.zeros((nch, n), np.int16)
    for c in range(nch):
        st = [int(rng.integers(25000, 50000)), int(rng.integers(75000, 95000))]
        pcm[c], _ = G.burst_oqpsk(n, burst_starts=st, ndata_sym=700, fc=8000.0 + rng.uniform(-60, 60), ebno_db=float(rng.uniform(10, 18)),
                                  seed=G.SEED_BASE + 500 + c)
    bank = bank_for(B, "burstoqpsk", {}, nch, capture_symbols=True, trace=True, max_write_samples=5000, softbit_capacity=30000)
    feed_random_chunks(bank, pcm, rng, 1, 5000, layout)
    naccepted = 0
    for c in list(range(0, nch, 9)) + [62, 64, 69]:
        ref = oracle_mod.run_burst(oracle_mod.burst_oqpsk_settings(), pcm[c], chunk=4096, capture_symbols=True, trace=True)
        check_soft(bank.read_softbits(c), ref["soft"], f"channel {c}")
        check_events(bank.read_events(c), ref["events"])
        sym = bank.read_symbols(c)
        assert sym.shape == ref["symbols"].shape
        assert np.max(np.abs(sym - ref["symbols"]), initial=0.0) < SYM_TOL
        naccepted += int((ref["soft"] == -1).sum())
    assert naccepted >= 10
    bank.close()


@pytest.mark.parametrize("fb", [1200, 600])
def test_burst_msk_bank_vs_oracle(B, oracle_mod, fb):
    nch = 5
    n = int(48000 * 4 * (1200 / fb))
    rng = np.random.default_rng(fb)
    pcm = np.zeros((nch, n), np.int16)
    for c in range(nch):
        pcm[c], _ = G.burst_msk(n, burst_starts=[int(n * rng.uniform(0.15, 0.3))], fb=float(fb), fc=1900.0 + rng.uniform(-300, 300),
                                ncw=int(rng.integers(112, 148)), ebno_db=float(rng.uniform(14, 22)), seed=G.SEED_BASE + 700 + c)
    opts = dict(fb=fb, lockingbw=1.5 * fb)
    bank = bank_for(B, "burstmsk", opts, nch, capture_symbols=True, trace=True, max_write_samples=8192, softbit_capacity=30000)
    feed_random_chunks(bank, pcm, rng, 100, 8192, 0)
    nacc = 0
    for c in range(nch):
        ref = oracle_mod.run_burst(oracle_settings(oracle_mod, "burstmsk", opts), pcm[c], chunk=4096, capture_symbols=True, trace=True)
        check_soft(bank.read_softbits(c), ref["soft"])
        check_events(bank.read_events(c), ref["events"])
        sym = bank.read_symbols(c)
        assert sym.shape == ref["symbols"].shape
        assert np.max(np.abs(sym - ref["symbols"]), initial=0.0) < SYM_TOL
        nacc += int((ref["soft"] == -1).sum())
    assert nacc >= 3
    bank.close()


def test_burst_oqpsk_squelch(B, oracle_mod):
    """setSQL(true): a group of 32 is emitted only if mse or the mse at the start of the write is under the threshold
    (burstoqpskdemodulator.cpp:318,708-715) -- write sizes matter, so the oracle is fed the same ones."""
    g = load_golden("burst_oqpsk_10k5_default")
    pcm = g["pcm"]
    bank = bank_for(B, "burstoqpsk", {}, 1, max_write_samples=3000, softbit_capacity=40000)
    bank.set_flags(afc=False, sql=True, cpu_reduce=False)
    for s in range(0, len(pcm), 3000):
        bank.write(pcm[None, s:s + 3000])
    ref = oracle_mod.run_burst(oracle_mod.burst_oqpsk_settings(), pcm, chunk=3000, sql=True)
    assert len(ref["soft"]) < len(g["soft"])  # something was squelched
    check_soft(bank.read_softbits(0), ref["soft"])
    bank.close()


def test_single_channel_mirror_groups(B):
    """BurstOqpskDemodulator mirror: processDemodulatedSoftBits is re-emitted in the reference's groups (marker + 32, then 32)."""
    g = load_golden("burst_oqpsk_10k5_default")
    d = B.BurstOqpskDemodulator(None, device=0, max_write_samples=4096, softbit_capacity=40000)
    groups, status = [], []
    d.processDemodulatedSoftBits = groups.append
    d.SignalStatus = status.append
    d.setSettings(B.BurstOqpskSettings())
    d.start()
    pcm = g["pcm"]
    for s in range(0, len(pcm), 4096):
        d.writeData(pcm[s:s + 4096].tobytes())
    flat = np.array([v for grp in groups for v in grp], dtype=np.int16)
    check_soft(flat, g["soft"])
    assert all(len(grp) in (32, 33) for grp in groups)
    assert status.count(True) == int((g["soft"] == -1).sum())


@pytest.mark.parametrize("afc,hz", [(False, 2300.0), (True, 1400.0)])
def test_burst_msk_center_freq_changed(B, oracle_mod, afc, hz):
    """BurstMskDemodulator::CenterFreqChangedSlot (burstmskdemodulator.cpp:327-342) on ONE channel of a live burst MSK bank between two
    writes: that channel equals the oracle given the same call at the same sample (the oracle = the unmodified reference there:
    tests/test_oracle_burst.py), its neighbours equal the oracle without it."""
    n, chunk, at = 48000 * 5, 4096, 90112
    pcm = np.stack([G.burst_msk(n, burst_starts=[30000 + 500 * c, 150000 + 300 * c], fb=1200.0, fc=1900.0 + 10 * c, ebno_db=18.0, seed=41 + c)[0]
                    for c in range(3)])
    bank = bank_for(B, "burstmsk", dict(fb=1200, lockingbw=1800.0), 3, max_write_samples=chunk, softbit_capacity=30000)
    bank.set_flags(afc=afc, sql=False, cpu_reduce=False)
    for s in range(0, n, chunk):
        if s == at:
            bank.center_freq_changed(hz, channel=1)
        bank.write(pcm[:, s:s + chunk])
    sett = oracle_mod.burst_msk_settings(fb=1200.0, lockingbw=1800.0)
    for c in range(3):
        ref = oracle_mod.run_burst(sett, pcm[c], chunk=chunk, afc=afc, center_at=at if c == 1 else -1, center_hz=hz)
        check_soft(bank.read_softbits(c), ref["soft"], f"channel {c}")
        check_events(bank.read_events(c), ref["events"])
        assert (ref["events"][:, 0] == at).any() == (c == 1)
        assert (ref["soft"] == -1).sum() >= 1
    bank.close()
    # the burst OQPSK slot is empty (burstoqpskdemodulator.cpp:284-289): accepted, nothing happens
    b2 = bank_for(B, "burstoqpsk", {}, 1, max_write_samples=chunk, softbit_capacity=1000)
    b2.center_freq_changed(7000.0)
    b2.close()


# ---------------------------------------------------------------------------------------------- setSettings on live burst channels
def oracle_with_sets(O, sett, pcm, sets, capture_symbols=True, trace=True):
    """The oracle object fed in pieces that end at the set points; sets = [(sample, Settings)] in ascending order."""
    d = O.BurstDemod(sett, capture_symbols=capture_symbols, trace=trace)
    s = 0
    for at, new in list(sets) + [(len(pcm), None)]:
        while s < at:
            e = min(s + 4096, at)
            d.write(pcm[s:e])
            s = e
        if new is not None:
            d.set_settings(new)
    out = {"soft": d.take_soft(), "events": d.take_events()}
    if capture_symbols:
        out["symbols"] = d.take_symbols()
    return out


def check_symbols_behind_sets(sym, ref, tag, nsets, sps):
    """SYM_TOL, except where the reference itself runs on rounding noise.  The restarted Hilbert filter's real part is -x[n - 1024]: exactly
    zero for the 1024 samples behind its latency (zero here too: the channel's history was cleared), but the reference's FFT leaves ~1e-17
    there, and the two AGCs that restart from empty averages (gain limit 1.4e6 each) turn that into symbols of up to 2e-5 (at most
    1024 / SamplesPerSymbol rows per call).  If a burst is running through the call, the reference's symbol-timing loop steers on
    arg(that noise) for those samples -- arbitrary angles, where the exact zeros here give arg = 0 -- and the rest of that burst's symbols
    come out ~1e-4 off (hard decisions equal, soft bytes within one: check_soft).  Everything else: SYM_TOL."""
    assert sym.shape == ref.shape, tag
    d = np.abs(sym - ref).max(axis=1)
    off = d >= SYM_TOL
    assert d.max(initial=0.0) < 2e-3, tag
    assert off.sum() <= nsets * (1024 / sps + 2) + 0.03 * len(ref), tag


def feed_with_sets(bank, pcm, rng, lo, hi, calls):
    """Ragged writes with a boundary at every call's sample; calls = [(sample, channel or -1, settings)] in ascending order."""
    s, n = 0, pcm.shape[1]
    calls = list(calls)
    while s < n:
        while calls and calls[0][0] == s:
            _, ch, new = calls.pop(0)
            bank.set_settings(new, channel=ch)
        m = min(int(rng.integers(lo, hi)), n - s)
        if calls:
            m = min(m, calls[0][0] - s)
        bank.write(pcm[:, s:s + m])
        s += m
    assert not calls


def oqpsk_set_case():
    nch, n = 70, 200000
    rng = np.random.default_rng(4242)
    pcm = np.zeros((nch, n), np.int16)
    starts = {}
    for c in range(nch):
        # the second burst more than a second behind the last call: setSettings replaces the AGC (a one-second moving average that starts from
        # zeros), and the reference loses most bursts that arrive before it has filled
        starts[c] = [int(rng.integers(25000, 40000)), int(rng.integers(140000, 155000))]
        pcm[c], _ = G.burst_oqpsk(n, burst_starts=starts[c], ndata_sym=700, fc=8000.0 + rng.uniform(-60, 60), ebno_db=float(rng.uniform(11, 18)),
                                  seed=G.SEED_BASE + 900 + c)
    per = {  # channel -> samples of its own calls
        1: [20011], 2: [starts[2][0] + 1500], 3: [starts[3][0] + 5000], 5: [starts[5][0] + 9000], 7: [starts[7][0] + 1000, starts[7][0] + 3000],
        63: [starts[63][0] + 4000], 64: [starts[64][0] + 2600], 69: [5, 17], 40: [starts[40][0] + 20000, starts[40][0] + 20009],
    }
    return pcm, per, 80001, sorted(set(per) | {0, 4, 6, 62, 65, 68}), rng


def test_burst_oqpsk_set_settings_live(B, oracle_mod):
    """BurstOqpskDemodulator::setSettings (burstoqpskdemodulator.cpp:202-277) on live channels of a 70-channel bank: single channels at their
    own moments (idle, between peak and trident check, inside the delayed burst, twice within one delay-line length), neighbours in the same
    wavefront untouched, then the whole bank at once -- every checked channel against the oracle given the same calls at the same samples
    (the oracle's live setSettings = the unmodified reference's: tests/test_oracle_burst.py).  With the trace on, the event log carries every
    firing of the peak detector and the metric of every trident check, so the check that runs 2633 samples behind each call -- on the ROTATED
    contents of d1 -- is compared too."""
    pcm, per, whole, check, rng = oqpsk_set_case()
    nch = pcm.shape[0]
    new = B.BurstOqpskSettings(freq_center=7000.0, lockingbw=9000.0, signalthreshold=0.55)
    onew = oracle_mod.burst_oqpsk_settings(freq_center=7000.0)
    onew.lockingbw, onew.signalthreshold = 9000.0, 0.55
    calls = sorted([(at, c, new) for c, ats in per.items() for at in ats] + [(whole, -1, new)], key=lambda x: x[0])
    bank = bank_for(B, "burstoqpsk", {}, nch, capture_symbols=True, trace=True, max_write_samples=5000, softbit_capacity=40000)
    feed_with_sets(bank, pcm, rng, 1, 5000, calls)
    nacc = 0
    for c in check:
        sets = sorted([(at, onew) for at in per.get(c, [])] + [(whole, onew)], key=lambda x: x[0])
        ref = oracle_with_sets(oracle_mod, oracle_mod.burst_oqpsk_settings(), pcm[c], sets)
        check_soft(bank.read_softbits(c), ref["soft"], f"channel {c}")
        check_events(bank.read_events(c), ref["events"])
        check_symbols_behind_sets(bank.read_symbols(c), ref["symbols"], c, len(sets), 2 * 48000 / 10500)
        nacc += int((ref["soft"] == -1).sum())
    assert nacc >= 15
    bank.close()


def msk_set_case(fb):
    nch = 6
    k = 1200 // fb
    n = 240000 * k
    rng = np.random.default_rng(fb + 1)
    pcm = np.zeros((nch, n), np.int16)
    starts = {}
    for c in range(nch):
        starts[c] = [int(k * rng.integers(28000, 36000)), int(k * rng.integers(150000, 160000))]
        pcm[c], _ = G.burst_msk(n, burst_starts=starts[c], fb=float(fb), fc=1900.0 + rng.uniform(-200, 200), ebno_db=float(rng.uniform(15, 22)),
                                seed=G.SEED_BASE + 950 + c)
    per = {0: [k * 20000 + 3], 1: [starts[1][0] + k * 3000], 2: [starts[2][0] + k * 15000], 3: [starts[3][0] + k * 30000, starts[3][0] + k * 34000],
           4: [starts[4][0] + k * 24000, starts[4][0] + k * 24011]}
    return pcm, per, k * 90000 + 1, rng


@pytest.mark.parametrize("fb", [1200, 600])
def test_burst_msk_set_settings_live(B, oracle_mod, fb):
    """BurstMskDemodulator::setSettings (burstmskdemodulator.cpp:150-325) on live channels: matched filters, AGCs, EbNo meter and resonator
    restart, cntr = 0 and mse = 10 in the middle of a burst, delayedsmpl keeps its contents with the pointer at zero."""
    pcm, per, whole, rng = msk_set_case(fb)
    nch = pcm.shape[0]
    opts = dict(fb=fb, lockingbw=1.5 * fb)
    new = B.BurstMskSettings(freq_center=1200.0, fb=float(fb), lockingbw=1.25 * fb, signalthreshold=0.55)
    onew = oracle_mod.burst_msk_settings(freq_center=1200.0, fb=float(fb), lockingbw=1.25 * fb)
    onew.signalthreshold = 0.55
    calls = sorted([(at, c, new) for c, ats in per.items() for at in ats] + [(whole, -1, new)], key=lambda x: x[0])
    bank = bank_for(B, "burstmsk", opts, nch, capture_symbols=True, trace=True, max_write_samples=8192, softbit_capacity=60000)
    feed_with_sets(bank, pcm, rng, 100, 8192, calls)
    nacc = 0
    for c in range(nch):
        sets = sorted([(at, onew) for at in per.get(c, [])] + [(whole, onew)], key=lambda x: x[0])
        ref = oracle_with_sets(oracle_mod, oracle_settings(oracle_mod, "burstmsk", opts), pcm[c], sets)
        check_soft(bank.read_softbits(c), ref["soft"], f"channel {c}")
        check_events(bank.read_events(c), ref["events"])
        check_symbols_behind_sets(bank.read_symbols(c), ref["symbols"], c, len(sets), 48000 / fb)
        nacc += int((ref["soft"] == -1).sum())
    assert nacc >= 6
    bank.close()


@pytest.mark.parametrize("fb0,fb1", [(1200, 600), (600, 1200)])
def test_burst_msk_live_rate_change(B, oracle_mod, fb0, fb1):
    """BurstMskDemodulator::setSettings with ANOTHER bit rate on the live object (the same class serves 600 and 1200 bps): every length changes;
    d1 / d2 / the peak detector's lines / delayedsmpl keep their first min(old, new) entries in storage order, startstop and the oscillator
    phases survive, the rest restarts.  jaero_set_settings re-creates the bank behind the handle with exactly those survivors (burst_rebank,
    k_burst_carry).  Four channels with staggered bursts, so the call falls into an idle stretch, between peak and trident check, and into the
    delayed burst; a burst at the new rate follows.  The oracle's live rate change is the unmodified reference's (checked in this file's CPU
    twin, tests/test_oracle_burst.py)."""
    nch, n = 4, 48000 * 9
    k0 = 1200 // fb0
    at = 70000 * k0 + 13
    rng = np.random.default_rng(fb0 + 7)
    pcm = np.zeros((nch, n), np.int16)
    for c in range(nch):
        first = at - k0 * [45000, 9000, 22000, 2000][c]
        a, _ = G.burst_msk(n, burst_starts=[first], fb=float(fb0), fc=1900.0 + 20 * c, ebno_db=18.0, seed=G.SEED_BASE + 970 + c)
        b, _ = G.burst_msk(n, burst_starts=[at + 150000 + 3000 * c], fb=float(fb1), fc=1880.0 + 20 * c, ebno_db=18.0, seed=G.SEED_BASE + 980 + c)
        cut = at + 60000
        pcm[c, :cut] = a[:cut]
        pcm[c, cut:] = b[cut:]
    opts0 = dict(fb=fb0, lockingbw=1.5 * fb0)
    new = B.BurstMskSettings(freq_center=1000.0, fb=float(fb1), lockingbw=1.5 * fb1)
    onew = oracle_mod.burst_msk_settings(freq_center=1000.0, fb=float(fb1), lockingbw=1.5 * fb1)
    bank = bank_for(B, "burstmsk", opts0, nch, capture_symbols=True, trace=True, max_write_samples=8192, softbit_capacity=60000)
    feed_with_sets(bank, pcm, rng, 100, 8192, [(at, -1, new)])
    nacc = 0
    for c in range(nch):
        ref = oracle_with_sets(oracle_mod, oracle_settings(oracle_mod, "burstmsk", opts0), pcm[c], [(at, onew)])
        check_soft(bank.read_softbits(c), ref["soft"], f"channel {c}")
        check_events(bank.read_events(c), ref["events"])
        check_symbols_behind_sets(bank.read_symbols(c), ref["symbols"], c, 1, 48000 / max(fb0, fb1))
        nacc += int((ref["soft"] == -1).sum())
    assert nacc >= nch  # every channel decodes its burst at the new rate
    bank.close()


def test_burst_set_settings_refusals(B):
    """Another kind is another class (JAERO_EINVAL); burst OQPSK exists at one rate; a bank's bit rate is shared by its channels."""
    from jaero_amd import capi

    bank = bank_for(B, "burstmsk", dict(fb=1200, lockingbw=1800.0), 2, max_write_samples=4096)
    with pytest.raises(capi.JaeroError):
        bank.set_settings(B.BurstOqpskSettings())
    with pytest.raises(capi.JaeroError):
        bank.set_settings(B.BurstMskSettings(fb=600.0, lockingbw=900.0), channel=1)
    bank.set_settings(B.BurstMskSettings(fb=1200.0, lockingbw=1500.0), channel=1)
    bank.close()
    b2 = bank_for(B, "burstoqpsk", {}, 1, max_write_samples=4096)
    with pytest.raises(capi.JaeroError):
        b2.set_settings(B.BurstOqpskSettings(fb=8400.0))
    b2.close()
