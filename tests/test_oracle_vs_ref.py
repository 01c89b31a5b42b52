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


@pytest.mark.parametrize("kind,seed,f0,e1,e2,chunk", [
    ("oqpsk", 51, (0, 0, 0), (31000, 1, 1, 1), (101000, 0, 0, 0), 1000),   # cpuReduce switched on mid-cycle, then off again
    ("oqpsk", 52, (1, 0, 1), (42000, 0, 1, 0), (113000, 1, 0, 1), 3000),   # starts reduced, AFC toggled with it
    ("oqpsk", 53, (0, 1, 1), (66000, 1, 0, 1), (67000, 0, 0, 0), 1000),    # two changes a thousand samples apart
    ("msk", 54, (0, 0, 0), (25000, 1, 0, 1), (90000, 0, 1, 0), 1000),
    ("msk", 55, (1, 1, 1), (33000, 0, 0, 0), (70500, 1, 0, 1), 1500),
])
def test_flags_changed_on_a_running_object(O, kind, seed, f0, e1, e2, chunk):
    """setAFC / setSQL / setCPUReduce called between two writes of a RUNNING object, at moments that are multiples neither of nfft/4 nor of 4096
    (oqpskdemodulator.cpp:149-163 + :399-431, mskdemodulator.cpp:105-118 + :345-368): the restatement against the unmodified reference.  What the GPU
    flags-matrix tests (tests/test_gpu_flags_matrix.py) compare with is the restatement fed such calls: here they are pinned.  cpuReduce changes the
    estimate cadence (ring fill gated by coarseCounter, nfft instead of nfft/4 between estimates) in the middle of a cycle."""
    if kind == "oqpsk":
        pcm, _ = G.oqpsk(170000, fc=8000 + 9.0 * (seed - 50), ebno_db=9.0 + seed % 4, seed=seed)
        st, kv = O.oqpsk_settings(), {}
    else:
        pcm, _ = G.msk(170000, fb=1200.0, fc=1000 + 4.0 * (seed - 50), ebno_db=11.0, seed=seed)
        st, kv = O.msk_settings(fb=1200.0, lockingbw=1800.0), dict(fb=1200, lockingbw=1800)
    ref = O.run_ref(kind, pcm, afc=f0[0], sql=f0[1], cpureduce=f0[2], chunk=chunk, flags_at=e1[0], flags_afc=e1[1], flags_sql=e1[2], flags_cpureduce=e1[3],
                    flags_at2=e2[0], flags2_afc=e2[1], flags2_sql=e2[2], flags2_cpureduce=e2[3], **kv)
    got = O.run_demod(st, pcm, afc=bool(f0[0]), sql=bool(f0[1]), cpu_reduce=bool(f0[2]), chunk=chunk,
                      flags_events=[(e1[0], bool(e1[1]), bool(e1[2]), bool(e1[3])), (e2[0], bool(e2[1]), bool(e2[2]), bool(e2[3]))])
    _cmp(ref, got)
    assert len(ref["status"]) >= 4  # estimates did fire on both sides of the changes


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


@pytest.mark.parametrize("kind,afc", [("oqpsk", 0), ("oqpsk", 1), ("msk", 1)])
def test_dcd_raised_and_dropped_on_a_running_object(O, kind, afc):
    """DCDstatSlot(true) and later DCDstatSlot(false) between writes (oqpskdemodulator.cpp:629-684, mskdemodulator.cpp:490-526: the flag steers
    FreqOffsetEstimateSlot's reset logic and, for MSK, the loop gains): restatement against the unmodified reference."""
    if kind == "oqpsk":
        pcm, _ = G.oqpsk(150000, fc=8013.0, ebno_db=10.0, seed=91 + afc)
        st, kv = O.oqpsk_settings(), {}
    else:
        pcm, _ = G.msk(150000, fb=1200.0, fc=1007.0, ebno_db=11.0, seed=93)
        st, kv = O.msk_settings(fb=1200.0, lockingbw=1800.0), dict(fb=1200, lockingbw=1800)
    ref = O.run_ref(kind, pcm, afc=afc, chunk=1000, dcd_at=37000, dcd_off_at=95000, **kv)
    got = O.run_demod(st, pcm, afc=bool(afc), chunk=1000, dcd_at=37000, dcd_off_at=95000)
    _cmp(ref, got)


@pytest.mark.parametrize("kind,set_at,chunk,cpu", [("oqpsk", 31000, 1000, 0), ("oqpsk", 70500, 1500, 1), ("msk", 25000, 1000, 0), ("msk", 41000, 1000, 1),
                                                   ("oqpsk8400", 41000, 1000, 0), ("oqpsk8400", 77000, 3500, 0)])
def test_same_rate_set_settings_at_an_unaligned_sample(O, kind, set_at, chunk, cpu):
    """setSettings with the SAME bit rate (another centre frequency / locking bandwidth) on a running object at a sample that is a multiple neither of nfft/4
    nor of 4096: the coarse ring pointer restarts there, so the object's estimates fire at other samples than its neighbours' from then on -- the case
    tests/test_gpu_flags_matrix.py builds inside one wavefront.  Restatement against the unmodified reference, with and without cpuReduce."""
    if kind == "oqpsk8400":
        # the C channel: setSettings also restarts the prefilter (fir_pre.SetKernel, :278-283) -- what jaero_set_settings does for ONE channel of an
        # 8400 bps bank since round 5 (tests/test_gpu_parity.py::test_8400_set_settings_on_one_channel_of_a_bank compares with the restatement)
        pcm, _ = G.oqpsk(150000, fb=8400.0, fc=8007.0, ebno_db=11.0, seed=85)
        st0, new = O.oqpsk_settings(fb=8400.0, lockingbw=8400.0), O.oqpsk_settings(fb=8400.0, lockingbw=7000.0, freq_center=8015.0)
        kv = dict(fb=8400, lockingbw=8400, set_freq_center=8015, set_lockingbw=7000)
        kind = "oqpsk"
    elif kind == "oqpsk":
        pcm, _ = G.oqpsk(220000 if cpu else 150000, fc=8011.0, ebno_db=11.0, seed=81 + cpu)  # (reduced: one estimate per 64 384 samples)
        st0, new = O.oqpsk_settings(), O.oqpsk_settings(freq_center=8025.0, lockingbw=9000.0)
        kv = dict(set_freq_center=8025, set_lockingbw=9000)
    else:
        pcm, _ = G.msk(220000 if cpu else 150000, fb=1200.0, fc=1006.0, ebno_db=11.0, seed=83 + cpu)
        st0, new = O.msk_settings(fb=1200.0, lockingbw=1800.0), O.msk_settings(fb=1200.0, lockingbw=1500.0, freq_center=1012.0)
        kv = dict(fb=1200, lockingbw=1800, set_freq_center=1012, set_lockingbw=1500)
    ref = O.run_ref(kind, pcm, cpureduce=cpu, chunk=chunk, set_at=set_at, **kv)
    got = O.run_demod(st0, pcm, cpu_reduce=bool(cpu), chunk=chunk, set_at=set_at, set_settings=new)
    _cmp(ref, got)
    assert len(ref["status"]) >= (2 if cpu else 3)  # estimates on both sides of the call


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
