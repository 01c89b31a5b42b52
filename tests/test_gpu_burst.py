"""GPU (-m gpu): the burst HIP path (Hilbert FIR -> front end -> trident FFT check -> tracking chain), called through the C ABI,
against the committed reference goldens and against the oracle on seeded multi-channel banks.

Contract (BASELINE.json north_star): hard decisions of the soft-bit stream bit-exact (incl. the -1 start-of-burst markers and
their positions), soft symbols within 1e-5; every SignalStatus / EbNo / Plottables emission at the same sample."""
import numpy as np
import pytest

from conftest import load_golden
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


def check_soft(got, ref, tag=""):
    assert len(got) == len(ref), tag
    assert np.array_equal(got == -1, ref == -1), "burst markers differ"
    assert np.array_equal(got >= 128, ref >= 128), "hard decisions differ"
    assert np.max(np.abs(got.astype(int) - ref.astype(int)), initial=0) <= 1


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
