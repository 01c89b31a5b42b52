"""CPU: the burst-demodulator restatement (oracle/jaero_oracle_burst.c) pinned against what the UNMODIFIED reference emitted
(tests/golden/burst_*.npz, made by oracle/_ref) and, where oracle/_ref can run, against the reference itself on fresh cases.
Bit-exact: soft bits (incl. the -1 start-of-burst markers) and every SignalStatus / EbNo / Plottables emission."""
import os

import numpy as np
import pytest

from conftest import load_golden
from jaero_amd import signalgen as G


def burst_settings(O, kind, opts):
    if kind == "burstoqpsk":
        return O.burst_oqpsk_settings(freq_center=opts.get("freq_center", 8000.0))
    fb = float(opts.get("fb", 1200))
    return O.burst_msk_settings(freq_center=opts.get("freq_center", 1000.0), fb=fb, lockingbw=opts.get("lockingbw", 1800.0))


def as_write_stamps(events, chunk):
    """The reference driver stamps an emission with the first sample of the write that carried it."""
    ev = events[events[:, 1] < 3].copy()
    ev[:, 0] = np.floor(ev[:, 0] / chunk) * chunk
    return ev


@pytest.mark.parametrize("name", ["burst_oqpsk_10k5_default", "burst_oqpsk_10k5_chunk1500", "burst_msk_1200_sample1_excerpt"])
def test_oracle_matches_reference_golden(oracle_mod, name):
    g = load_golden(name)
    chunk = g["opts"].get("chunk", 4096)
    o = oracle_mod.run_burst(burst_settings(oracle_mod, g["kind"], g["opts"]), g["pcm"], chunk=chunk)
    assert np.array_equal(o["soft"], g["soft"])
    assert (g["soft"] == -1).sum() >= 2  # the fixtures hold at least two accepted bursts
    assert np.array_equal(as_write_stamps(o["events"], chunk), g["events"])


def test_generator_bits_come_back(oracle_mod):
    """The burst OQPSK fixture decodes: hard decisions of the two output streams equal the transmitted arm bits (up to the
    per-burst 4-fold ambiguity) -- guards the generator and the marker position, not just self-consistency."""
    g = load_golden("burst_oqpsk_10k5_default")
    soft = g["soft"]
    marks = np.nonzero(soft == -1)[0]
    tx = g["tx_bits"].reshape(len(g["tx_starts"]), -1)
    ok = 0
    for b, m in enumerate(marks[-len(tx):]):
        end = marks[marks > m][0] if (marks > m).any() else len(soft)
        seg = soft[m + 1:end]
        hard = (seg >= 128).astype(np.uint8)
        s0, s1 = hard[0::2], hard[1::2]
        best = 1.0
        for arm in (tx[b][0::2], tx[b][1::2]):
            for stream in (s0, s1):
                for lag in range(0, 40):
                    n = min(len(arm), len(stream) - lag, 600)
                    if n < 300:
                        continue
                    e = np.mean(arm[:n] != stream[lag:lag + n])
                    best = min(best, e, 1 - e)
        ok += best < 0.02
    assert ok >= 1


def test_chunk_invariance(oracle_mod):
    g = load_golden("burst_oqpsk_10k5_default")
    a = oracle_mod.run_burst(oracle_mod.burst_oqpsk_settings(), g["pcm"], chunk=4096)
    b = oracle_mod.run_burst(oracle_mod.burst_oqpsk_settings(), g["pcm"], chunk=777)
    assert np.array_equal(a["soft"], b["soft"]) and np.array_equal(a["events"], b["events"])


def test_hilbert_is_a_delayed_fir(oracle_mod):
    """QJHilbertFilter through JFastFir == causal convolution with the 2048-tap kernel delayed by nfft-K+1 samples
    (the property JAERO/tests/jfastfir_tests.cpp pins for JFastFir), independent of the write sizes."""
    rng = np.random.default_rng(5)
    x = rng.integers(-20000, 20000, size=30000).astype(np.int16)
    y, lat = oracle_mod.hilbert_stream(x, chunk=1000)
    y2, _ = oracle_mod.hilbert_stream(x, chunk=4096)
    assert np.allclose(y, y2, atol=1e-12)
    k = oracle_mod.hilbert_kernel(2048)
    full = np.convolve(x.astype(np.float64) / 32768.0, k)[: len(x)]
    ref = np.concatenate([np.zeros(lat, complex), full])[: len(x)]
    assert lat == 8192 - 2048 + 1
    assert np.max(np.abs(y - ref)) < 1e-12


# ------------------------------------------------------------------------------------------- against _ref itself
@pytest.fixture(scope="module")
def R(oracle_mod):
    if not oracle_mod.have_ref():
        pytest.skip("oracle/_ref/jaero_ref not available here")
    try:
        oracle_mod.run_ref("burstmsk", np.zeros(16, np.int16))
    except Exception as e:
        pytest.skip(f"_ref cannot run here: {e}")
    return oracle_mod


@pytest.mark.parametrize("fb,seed,chunk", [(1200, 31, 4096), (600, 32, 3000)])
def test_burst_msk_synthetic_vs_ref(R, fb, seed, chunk):
    n = int(48000 * 4 * (1200 / fb))
    pcm, _ = G.burst_msk(n, burst_starts=[n // 5], fb=float(fb), fc=1900.0 + seed, ebno_db=18.0, seed=seed)
    r = R.run_ref("burstmsk", pcm, fb=fb, lockingbw=1.5 * fb, chunk=chunk)
    o = R.run_burst(R.burst_msk_settings(fb=float(fb), lockingbw=1.5 * fb), pcm, chunk=chunk)
    assert (r["soft"] == -1).sum() >= 1
    assert np.array_equal(r["soft"], o["soft"])
    assert np.array_equal(r["events"], as_write_stamps(o["events"], chunk))


@pytest.mark.parametrize("afc,hz", [(False, 2300.0), (True, 1400.0), (False, 100.0)])
def test_burst_msk_center_freq_changed_vs_ref(R, afc, hz):
    """BurstMskDemodulator::CenterFreqChangedSlot in the middle of a stream (burstmskdemodulator.cpp:327-342): the clamp to
    [0.75 fb, Fs/2 - 0.75 fb], mixer2 following under AFC or pulled to within lockingbw/2, the Plottables emission -- the unmodified
    reference (center_at / center_hz of oracle/ref/driver.cpp) against the restatement."""
    n = 48000 * 5
    pcm, _ = G.burst_msk(n, burst_starts=[30000, 150000], fb=1200.0, fc=1900.0, ebno_db=18.0, seed=41)
    r = R.run_ref("burstmsk", pcm, fb=1200, lockingbw=1800, chunk=4096, center_at=90112, center_hz=hz, afc=int(afc))
    o = R.run_burst(R.burst_msk_settings(fb=1200.0, lockingbw=1800.0), pcm, chunk=4096, afc=afc, center_at=90112, center_hz=hz)
    assert np.array_equal(r["soft"], o["soft"])
    assert np.array_equal(r["events"], as_write_stamps(o["events"], 4096))
    assert (o["events"][:, 0] == 90112).any()  # the slot's own Plottables emission


def test_burst_oqpsk_random_vs_ref(R):
    pcm, _ = G.burst_oqpsk(100000, burst_starts=[30000], ndata_sym=800, fc=7990.0, ebno_db=12.0, seed=77)
    r = R.run_ref("burstoqpsk", pcm, chunk=2000)
    o = R.run_burst(R.burst_oqpsk_settings(), pcm, chunk=2000)
    assert np.array_equal(r["soft"], o["soft"])
    assert np.array_equal(r["events"], as_write_stamps(o["events"], 2000))


@pytest.mark.parametrize("f", ["1200bps_burst_sample1", "1200bps_burst_sample2"])
def test_bundled_recordings(oracle_mod, f):
    """The reference's bundled 1200 bps burst recordings through the burst MSK demodulator (inputs stay in /root/reference)."""
    path = f"/root/reference/samples/{f}.wav"
    if not os.path.exists(path):
        pytest.skip("reference samples not present on this machine")
    import wave

    w = wave.open(path)
    x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
    g = load_golden(f + "_burstmsk")
    assert int(g["nsamples"]) == len(x)
    o = oracle_mod.run_burst(oracle_mod.burst_msk_settings(), x, chunk=4096)
    assert np.array_equal(o["soft"], g["soft"])
    assert np.array_equal(as_write_stamps(o["events"], 4096), g["events"])


# setSettings on the live object (a user pressing OK in the settings dialog while audio runs): what survives is a member-by-member mixture --
# new AGCs / EbNo meter / moving averages, Hilbert filter, peak detector and trident fill restarted, but DelayThing::setLength (DSP.h:447-453)
# keeps the old CONTENTS of d1 / d2 with the pointer back at zero, startstop / mse / oscillator phases stay (burst OQPSK), cntr and the matched
# filters restart (burst MSK).  Positions: idle, between the peak detector's firing and the trident check, while the delayed burst runs
# through the demodulator, and twice in a row (the second call inside the first one's transient).
SET_CASES_OQPSK = [(25000, -1), (31500, -1), (35000, -1), (40000, -1), (31000, 33000), (52000, 53000)]


@pytest.mark.parametrize("set_at,set_at2", SET_CASES_OQPSK)
def test_burst_oqpsk_set_settings_vs_ref(R, set_at, set_at2):
    pcm, _ = G.burst_oqpsk(130000, burst_starts=[30000, 80000], ndata_sym=800, fc=7990.0, ebno_db=14.0, seed=91)
    kw = dict(set_at=set_at, set_lockingbw=9000, set_freq_center=7000, set_threshold=0.55)
    if set_at2 >= 0:
        kw["set_at2"] = set_at2
    r = R.run_ref("burstoqpsk", pcm, chunk=1000, **kw)
    new = R.burst_oqpsk_settings(freq_center=7000.0)
    new.lockingbw, new.signalthreshold = 9000.0, 0.55
    o = R.run_burst(R.burst_oqpsk_settings(), pcm, chunk=1000, set_at=[a for a in (set_at, set_at2) if a >= 0], set_settings=new)
    assert np.array_equal(r["soft"], o["soft"])
    assert np.array_equal(r["events"], as_write_stamps(o["events"], 1000))
    assert (o["events"][:, 0] == -(-set_at // 1000) * 1000).any()  # setSettings' own Plottables emission, in front of the first write at or behind set_at
    assert (o["soft"] == -1).sum() >= 1


@pytest.mark.parametrize("fb,set_at,set_at2", [(1200, 20000, -1), (1200, 33000, -1), (1200, 45000, -1), (1200, 60000, 64000), (600, 50000, -1), (600, 90000, 100000)])
def test_burst_msk_set_settings_vs_ref(R, fb, set_at, set_at2):
    n = int(48000 * 5 * (1200 / fb))
    starts = [30000, 130000] if fb == 1200 else [40000, 260000]
    pcm, _ = G.burst_msk(n, burst_starts=starts, fb=float(fb), fc=1900.0, ebno_db=18.0, seed=17 + fb)
    kw = dict(set_at=set_at, set_lockingbw=1500, set_freq_center=1200, set_threshold=0.55)
    if set_at2 >= 0:
        kw["set_at2"] = set_at2
    r = R.run_ref("burstmsk", pcm, fb=fb, lockingbw=1800, chunk=1000, **kw)
    new = R.burst_msk_settings(freq_center=1200.0, fb=float(fb), lockingbw=1500.0)
    new.signalthreshold = 0.55
    o = R.run_burst(R.burst_msk_settings(fb=float(fb), lockingbw=1800.0), pcm, chunk=1000, set_at=[a for a in (set_at, set_at2) if a >= 0], set_settings=new)
    assert np.array_equal(r["soft"], o["soft"])
    assert np.array_equal(r["events"], as_write_stamps(o["events"], 1000))
    assert (o["events"][:, 0] == -(-set_at // 1000) * 1000).any()


@pytest.mark.parametrize("fb0,fb1,set_at", [(1200, 600, 60000), (600, 1200, 100000), (1200, 600, 45000), (600, 1200, 61000)])
def test_burst_msk_live_rate_change_vs_ref(R, fb0, fb1, set_at):
    """The same BurstMskDemodulator object serves 600 and 1200 bps: setSettings with the other rate on the live object resizes every DelayThing
    (the first min(old, new) entries stay, in storage order) and restarts the rest.  A burst at the old rate in front of the call (the call
    falls behind it, into it, or in front of its trident check), one at the new rate behind it: reference against restatement, bit-exact."""
    n = 48000 * 8
    a, _ = G.burst_msk(n, burst_starts=[30000], fb=float(fb0), fc=1900.0, ebno_db=18.0, seed=5)
    b, _ = G.burst_msk(n, burst_starts=[200000], fb=float(fb1), fc=1900.0, ebno_db=18.0, seed=6)
    pcm = a.copy()
    pcm[150000:] = b[150000:]
    r = R.run_ref("burstmsk", pcm, fb=fb0, lockingbw=1.5 * fb0, chunk=1000, set_at=set_at, set_fb=fb1, set_lockingbw=1.5 * fb1)
    new = R.burst_msk_settings(fb=float(fb1), lockingbw=1.5 * fb1)
    o = R.run_burst(R.burst_msk_settings(fb=float(fb0), lockingbw=1.5 * fb0), pcm, chunk=1000, set_at=[set_at], set_settings=new)
    assert np.array_equal(r["soft"], o["soft"])
    assert np.array_equal(r["events"], as_write_stamps(o["events"], 1000))
    assert (o["soft"] == -1).sum() >= 1 and (o["events"][:, 0] > 200000).any()  # the burst at the new rate is found
