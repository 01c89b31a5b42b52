"""CPU, build container only: the C restatement against the unmodified reference binary (oracle/_ref) on fresh
random cases.  Skipped where _ref cannot run (the GPU box has no /root/reference; it uses the committed fixtures)."""
import numpy as np
import pytest

from jaero_amd import signalgen as G


def _cmp(r, o):
    assert np.array_equal(r["soft"], o["soft"])
    assert r["status"].shape == o["status"].shape
    assert np.array_equal(r["status"][:, [0, 1, 2, 3, 5]], o["status"][:, [0, 1, 2, 3, 5]])


@pytest.fixture(scope="module")
def O(oracle_mod):
    if not oracle_mod.have_ref():
        pytest.skip("oracle/_ref/jaero_ref not available here")
    try:
        oracle_mod.run_ref("msk", np.zeros(16, np.int16))
    except Exception as e:  # Qt runtime missing on this machine
        pytest.skip(f"_ref cannot run here: {e}")
    return oracle_mod


@pytest.mark.parametrize("seed,afc,cpu,chunk", [(11, 0, 0, 4096), (12, 1, 0, 1500), (13, 0, 1, 4096)])
def test_oqpsk_random(O, seed, afc, cpu, chunk):
    pcm, _ = G.oqpsk(60000, fc=8000 + 7.0 * seed, ebno_db=8.0 + seed % 5, seed=seed)
    _cmp(O.run_ref("oqpsk", pcm, afc=afc, cpureduce=cpu, chunk=chunk),
         O.run_demod(O.oqpsk_settings(), pcm, afc=bool(afc), cpu_reduce=bool(cpu), chunk=chunk))


@pytest.mark.parametrize("seed,afc,cpu,chunk,center", [(31, 0, 0, 4096, False), (32, 1, 0, 1000, False), (33, 0, 1, 4096, False), (34, 0, 0, 2500, True)])
def test_oqpsk_8400_random(O, seed, afc, cpu, chunk, center):
    """fb == 8400: JFastFir prefilter with its per-write mixer update, centre-weighted coarse window, alpha-0.6 filters, the
    carrier-loop gains of the <= 8400 branch (oqpskdemodulator.cpp:345-381,436-448,518-532,607-608, coarsefreqestimate.cpp:100)."""
    pcm, _ = G.oqpsk(130000, fb=8400.0, fc=8000 + 3.0 * seed, ebno_db=8.0 + seed % 5, seed=seed)
    kv = dict(center_at=60000, center_hz=8090) if center else {}
    _cmp(O.run_ref("oqpsk", pcm, fb=8400, lockingbw=8400, afc=afc, cpureduce=cpu, chunk=chunk, dcd_at=100000, **kv),
         O.run_demod(O.oqpsk_settings(fb=8400.0, lockingbw=8400.0), pcm, afc=bool(afc), cpu_reduce=bool(cpu), chunk=chunk, dcd_at=100000, **kv))


@pytest.mark.parametrize("fb,seed", [(1200, 21), (600, 22)])
def test_msk_random(O, fb, seed):
    pcm, _ = G.msk(60000, fb=fb, fc=1000 + seed, ebno_db=11.0, seed=seed)
    bw = 1800 if fb == 1200 else 900
    _cmp(O.run_ref("msk", pcm, fb=fb, lockingbw=bw, dcd_at=20000),
         O.run_demod(O.msk_settings(fb=fb, lockingbw=bw), pcm, dcd_at=20000))


@pytest.mark.parametrize("Fs,fb", [(24000, 1200), (24000, 600), (12000, 1200), (12000, 600)])
def test_msk_other_sample_rates(O, Fs, fb):
    """What MskDemodulator::dataReceived does when audio arrives at another rate (mskdemodulator.cpp:528-537): setSettings with that Fs."""
    pcm, _ = G.msk(int(Fs * 6), fb=float(fb), Fs=float(Fs), fc=1004.0, ebno_db=11.0, seed=40 + fb // 600 + Fs // 12000)
    _cmp(O.run_ref("msk", pcm, fb=fb, lockingbw=1.5 * fb, Fs=Fs, chunk=3000),
         O.run_demod(O.msk_settings(fb=float(fb), lockingbw=1.5 * fb, Fs=float(Fs)), pcm, chunk=3000))


def test_noise_only_and_center_change(O):
    rng = np.random.default_rng(3)
    noise = rng.normal(0, 2500, 60000).astype(np.int16)
    _cmp(O.run_ref("oqpsk", noise, center_at=30000, center_hz=8100),
         O.run_demod(O.oqpsk_settings(), noise, center_at=30000, center_hz=8100))


def test_fft_shim_matches_restatement(O):
    rng = np.random.default_rng(4)
    x = rng.normal(size=256) + 1j * rng.normal(size=256)
    y = O.ref_tool("fft", x.astype(np.complex128), np.complex128, n=256)
    z = x.astype(np.complex128).copy()
    O.lib().jo_fft(z.ctypes.data, 256, 0)
    assert np.array_equal(y, z)


@pytest.mark.parametrize("fb0,fb1,set_at", [(8400, 10500, 20480), (10500, 8400, 24576), (8400, 8400, 20480)])
def test_oqpsk_live_rate_change(O, fb0, fb1, set_at):
    """OqpskDemodulator::setSettings on a running object with another bit rate (oqpskdemodulator.cpp:175-289: AGC, filters, delays and
    resonator are rebuilt, oscillator phases, moving averages, the smoothed spectrum and the loop filter stay): the restatement against
    the reference, which received the same call between the same two writes."""
    pcm, _ = G.oqpsk(90000, fc=8012.0, ebno_db=12.0, seed=71, fb=10500.0)
    new = O.oqpsk_settings(fb=float(fb1), lockingbw=float(fb1))
    _cmp(O.run_ref("oqpsk", pcm, fb=fb0, lockingbw=fb0, set_at=set_at, set_fb=fb1, set_lockingbw=fb1),
         O.run_demod(O.oqpsk_settings(fb=float(fb0), lockingbw=float(fb0)), pcm, set_at=set_at, set_settings=new))


@pytest.mark.parametrize("Fs0,fb0,Fs1,fb1", [(48000, 600, 48000, 1200), (48000, 1200, 24000, 1200), (24000, 600, 48000, 600)])
def test_msk_live_rate_change(O, Fs0, fb0, Fs1, fb1):
    """MskDemodulator::setSettings with another bit rate / sample rate on a running object (mskdemodulator.cpp:135-263; what dataReceived
    does when audio arrives at another rate, :528-537)."""
    pcm, _ = G.msk(int(Fs1 * 3), fb=float(fb1), Fs=float(Fs1), fc=1004.0, ebno_db=12.0, seed=77)
    set_at = 9000
    new = O.msk_settings(fb=float(fb1), lockingbw=1.5 * fb1, Fs=float(Fs1))
    _cmp(O.run_ref("msk", pcm, fb=fb0, lockingbw=1.5 * fb0, Fs=Fs0, chunk=3000, set_at=set_at, set_fb=fb1, set_Fs=Fs1, set_lockingbw=1.5 * fb1),
         O.run_demod(O.msk_settings(fb=float(fb0), lockingbw=1.5 * fb0, Fs=float(Fs0)), pcm, chunk=3000, set_at=set_at, set_settings=new))


@pytest.mark.parametrize("afc,cpu,chunk,dcd_at,center", [(1, 0, 1000, -1, False), (0, 1, 4096, -1, False), (0, 0, 777, 200000, False), (1, 0, 4096, -1, True)])
def test_oqpsk_on_the_sample_recording(O, afc, cpu, chunk, dcd_at, center):
    """The reference's own 10.5 kbps recording (tests/golden/recording_oqpsk_10k5.npz: 12 s of samples/10.5k_sample.ogg) with the options the
    synthetic cases use -- AFC, CPU reduction, odd write sizes, a DCD change, the user moving the centre frequency onto the carrier: reference
    against restatement, every soft bit and status row."""
    from conftest import load_golden

    pcm = load_golden("recording_oqpsk_10k5")["pcm"]
    kv = dict(center_at=150000, center_hz=5760) if center else {}
    if dcd_at >= 0:
        kv["dcd_at"] = dcd_at
    r = O.run_ref("oqpsk", pcm, afc=afc, cpureduce=cpu, chunk=chunk, **kv)
    o = O.run_demod(O.oqpsk_settings(), pcm, afc=bool(afc), cpu_reduce=bool(cpu), chunk=chunk, **kv)
    _cmp(r, o)
    assert len(r["soft"]) > 100000 and r["status"][-1, 5] == 1
