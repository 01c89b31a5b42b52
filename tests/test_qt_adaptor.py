"""The Qt adaptors of integration/qt/hipdemodulator.h (HipOqpskDemodulator / HipMskDemodulator: QIODevice subclasses over the C ABI)
compiled with moc against Qt 5.9.7 and wired, as MainWindow wires the reference's own classes, to the UNMODIFIED AeroL:
what AeroL prints (signal units, DCD changes) must be the same text with either demodulator in front of it.

oracle/_ref/adaptor_demo is built by `make -C oracle adaptor` (it links reference objects, so only where /root/reference exists; the
binary travels to the GPU box like oracle/_ref/jaero_ref).  The CPU test checks that it exists, runs its reference side and exports
the expected surface; the GPU test runs both sides."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from jaero_amd import aerol_frames as AF
from jaero_amd import signalgen as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "oracle", "_ref", "adaptor_demo")
have_demo = pytest.mark.skipif(not os.path.exists(DEMO), reason="oracle/_ref/adaptor_demo not built (needs /root/reference + Qt: make -C oracle adaptor)")


def p_channel_pcm(nfr=8, fc=8011.0, seed=5):
    pay = AF.random_payloads(nfr, 10500, seed=seed)
    bits, _ = AF.p_channel_bits(pay, 10500)
    n = int(len(bits) / 2 * 48000 / 5250) + 2000
    pcm, _ = G.oqpsk(n, fc=fc, ebno_db=13.0, seed=seed + 20, bits=np.concatenate([bits, np.zeros(64, np.uint8)]))
    return pcm, pay


def run_demo(impl, kind, pcm, **kv):
    env = dict(os.environ, QT_QPA_PLATFORM="offscreen")
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "in.s16"), os.path.join(td, "out.txt")
        np.ascontiguousarray(pcm, dtype=np.int16).tofile(inp)
        subprocess.check_call([DEMO, impl, kind, inp, outp] + [f"{k}={v}" for k, v in kv.items()], env=env, stdout=subprocess.DEVNULL)
        return open(outp, "rb").read().decode("latin1")


@have_demo
def test_reference_side_decodes_the_frames():
    """Sanity of the driver itself: the all-reference chain prints the transmitted signal units."""
    import re

    pcm, pay = p_channel_pcm(nfr=14)
    txt = run_demo("ref", "oqpsk", pcm)
    assert "#DCD 1" in txt
    sent = {"".join("%02X" % b for b in p) for fr in pay for p in fr}
    good = [ln for ln in txt.split("\n") if re.match(r"^. 0x", ln) and "Bad CRC" not in ln]  # the first frames arrive while the AGC settles
    got = ["".join(re.findall(r"0x([0-9A-F]{2})", ln)) for ln in good]
    assert len(got) >= 26 * 6 and all(g in sent for g in got)


def test_adaptor_header_mirrors_the_reference_surface():
    """Every member MainWindow uses on the reference classes exists in the adaptor header (names as in oqpskdemodulator.h:41-67)."""
    src = open(os.path.join(ROOT, "integration", "qt", "hipdemodulator.h")).read()
    for name in ("HipBurstOqpskDemodulator", "HipBurstMskDemodulator", "setScatterPointType", "invalidatesettings"):
        assert name in src, name
    for name in ("setSettings", "setAFC", "setSQL", "setCPUReduce", "start", "stop", "getCurrentFreq", "writeData", "readData",
                 "processDemodulatedSoftBits", "Plottables", "MSESignal", "SignalStatus", "EbNoMeasurmentSignal", "SampleRateChanged",
                 "BitRateChanged", "CenterFreqChangedSlot", "DCDstatSlot", "dataReceived", "WarningTextSignal"):
        assert name in src, name


@pytest.mark.gpu
@have_demo
def test_hip_adaptor_under_unmodified_aerol_oqpsk():
    pcm, _ = p_channel_pcm(nfr=14)
    ref = run_demo("ref", "oqpsk", pcm)
    hip = run_demo("hip", "oqpsk", pcm)
    assert ref.count("#DCD 1") >= 1 and len(ref) > 2000
    assert hip == ref


@pytest.mark.gpu
@have_demo
def test_hip_adaptor_rate_change_carries_state_over():
    """A user changing the bit rate: setSettings(8400) + some input, then setSettings(10500).  The reference rebuilds AGC, filters, delays
    and windows inside the old object but keeps its oscillator phases, loop states and symbol-rate window contents (which lets it decode
    the first frames a little earlier than a fresh demodulator: 260 against 234 CRC-clean lines on this input).  jaero_set_settings
    re-creates the one-channel bank behind the handle with exactly those survivors (rebank_with_carry_over), so the adaptor under the
    unmodified AeroL must print what the all-reference chain prints for the same episode."""
    pcm, _ = p_channel_pcm(nfr=14)
    ref = run_demo("ref", "oqpsk", pcm, prefb=8400)
    hip = run_demo("hip", "oqpsk", pcm, prefb=8400)
    fresh = run_demo("ref", "oqpsk", pcm)
    assert len(ref) > 2000 and ref != fresh  # the episode leaves a trace in the reference's output ...
    assert hip == ref                        # ... and the same trace here


@pytest.mark.gpu
@have_demo
def test_hip_adaptor_under_unmodified_aerol_msk():
    pay = AF.random_payloads(8, 1200, seed=9)
    bits, _ = AF.p_channel_bits(pay, 1200)
    n = int(len(bits) * 48000 / 1200) + 4000
    pcm, _ = G.msk(n, fb=1200.0, fc=1007.0, ebno_db=16.0, seed=31, bits=np.concatenate([bits, np.zeros(16, np.uint8)]))
    ref = run_demo("ref", "msk", pcm, fb=1200)
    hip = run_demo("hip", "msk", pcm, fb=1200)
    assert len(ref) > 500
    assert hip == ref


def rt_burst_pcm():
    """Three R / T packets as 10.5 kbps burst OQPSK passband PCM (as tests/test_gpu_aerol_burst.py builds them)."""
    n = 48000 * 5
    rng = np.random.default_rng(21)
    rb = lambda k: bytes(rng.integers(0, 256, k, dtype=np.uint8))
    uw = np.repeat(np.array([(AF.UW >> (31 - k)) & 1 for k in range(32)], dtype=np.uint8), 2)
    pk = [("R", rb(17)), ("T", (rb(4), [rb(10) for _ in range(4)])), ("R", rb(17))]
    data = [np.concatenate([uw, AF.rt_packet_bits(k, p)]) for k, p in pk]
    pcm, _ = G.burst_oqpsk(n, burst_starts=[40000, 100000, 185000], ndata_sym=900, fc=8015.0, ebno_db=16.0, seed=G.SEED_BASE + 300, data=data)
    return pcm


@have_demo
def test_reference_side_decodes_the_bursts():
    """Sanity of the burst half of the driver: BurstOqpskDemodulator -> AeroL (burst mode), all reference, prints R / T packets."""
    txt = run_demo("ref", "burstoqpsk", rt_burst_pcm())
    assert "T Packet from AES" in txt and "R_channel" in txt and "#DCD 1" in txt


@pytest.mark.gpu
@have_demo
def test_hip_burst_adaptor_under_unmodified_aerol_oqpsk():
    """HipBurstOqpskDemodulator (soft-bit groups cut as the reference cuts them, -1 start-of-burst marker) under the unmodified AeroL in
    burst mode: the same text as the all-reference chain."""
    pcm = rt_burst_pcm()
    ref = run_demo("ref", "burstoqpsk", pcm)
    hip = run_demo("hip", "burstoqpsk", pcm)
    assert len(ref) > 100
    assert hip == ref


@pytest.mark.gpu
@have_demo
@pytest.mark.parametrize("fb", [1200, 600])
def test_hip_burst_adaptor_groups_msk(fb):
    """HipBurstMskDemodulator: the soft-bit groups it hands to AeroL (dumped by the driver: sizes 12 / 13 with the -1 marker, values)
    are the groups BurstMskDemodulator hands over, and AeroL prints the same lines (DCD changes feed back through DCDstatSlot)."""
    pcm, bursts = G.burst_msk(48000 * 6 if fb == 1200 else 48000 * 9, burst_starts=[30000, 150000] if fb == 1200 else [30000, 230000],
                              ndata=400, fb=float(fb), fc=1007.0, ebno_db=18.0, seed=G.SEED_BASE + 77 + fb)
    ref = run_demo("ref", "burstmsk", pcm, fb=fb, dump=1)
    hip = run_demo("hip", "burstmsk", pcm, fb=fb, dump=1)
    groups = [ln for ln in ref.split("\n") if ln.startswith("G ")]
    assert len(groups) >= 40 and any(" -1" in ln for ln in groups)
    assert hip == ref


@pytest.mark.gpu
@have_demo
@pytest.mark.parametrize("set_at", [70000, 41500])
def test_hip_burst_adaptor_live_set_settings_oqpsk(set_at):
    """A user pressing OK in the settings dialog while burst audio runs: BurstOqpskDemodulator::setSettings on the live object
    (burstoqpskdemodulator.cpp:202-277).  The adaptor passes it to jaero_set_settings (k_burst_settings.h) instead of replacing its bank;
    under the unmodified AeroL it prints what the all-reference chain prints for the same episode -- between two bursts (the later packets
    decode, the trident check that runs 2633 samples behind the call sees the rotated contents of d1), and right behind the first burst's
    peak (that packet is lost on both sides)."""
    pcm = rt_burst_pcm()
    kw = dict(set_at=set_at, set_lockingbw=9000, set_freq_center=7900)
    ref = run_demo("ref", "burstoqpsk", pcm, **kw)
    hip = run_demo("hip", "burstoqpsk", pcm, **kw)
    plain = run_demo("ref", "burstoqpsk", pcm)
    assert len(ref) > 100 and ref != plain  # the call leaves a trace in the reference's output ...
    assert hip == ref                       # ... and the same one here


@pytest.mark.gpu
@have_demo
def test_hip_burst_adaptor_live_set_settings_msk():
    """The same for BurstMskDemodulator::setSettings (burstmskdemodulator.cpp:150-325) in the middle of the first burst: cntr = 0 and
    mse = 10 with startstop still counting, matched filters and AGCs empty, delayedsmpl rotated; the groups AeroL receives are compared."""
    pcm, _ = G.burst_msk(48000 * 6, burst_starts=[30000, 170000], ndata=400, fb=1200.0, fc=1007.0, ebno_db=18.0, seed=G.SEED_BASE + 79)
    kw = dict(fb=1200, dump=1, set_at=60000, set_lockingbw=1500, set_freq_center=1100)
    ref = run_demo("ref", "burstmsk", pcm, **kw)
    hip = run_demo("hip", "burstmsk", pcm, **kw)
    plain = run_demo("ref", "burstmsk", pcm, fb=1200, dump=1)
    assert len([ln for ln in ref.split("\n") if ln.startswith("G ")]) >= 20 and ref != plain
    assert hip == ref


@pytest.mark.gpu
@have_demo
def test_hip_burst_adaptor_live_rate_change_msk():
    """Burst 1200 -> burst 600 on the running demodulator (the same BurstMskDemodulator object serves both): the adaptor hands the call to
    jaero_set_settings, which re-creates its bank with what the reference keeps; the groups AeroL receives for a 600 bps burst that follows
    are the all-reference chain's."""
    a, _ = G.burst_msk(48000 * 9, burst_starts=[30000], ndata=300, fb=1200.0, fc=1007.0, ebno_db=18.0, seed=G.SEED_BASE + 81)
    b, _ = G.burst_msk(48000 * 9, burst_starts=[200000], ndata=300, fb=600.0, fc=1007.0, ebno_db=18.0, seed=G.SEED_BASE + 82)
    pcm = a.copy()
    pcm[150000:] = b[150000:]
    kw = dict(fb=1200, dump=1, set_at=100000, set_fb=600, set_lockingbw=900)
    ref = run_demo("ref", "burstmsk", pcm, **kw)
    hip = run_demo("hip", "burstmsk", pcm, **kw)
    groups = [ln for ln in ref.split("\n") if ln.startswith("G ")]
    assert len(groups) >= 40 and sum(" -1" in ln for ln in groups) >= 2  # a burst at each rate
    assert hip == ref


@pytest.mark.gpu
@have_demo
def test_hip_msk_adaptor_follows_the_incoming_sample_rate():
    """Audio arriving through dataReceived at 24 kHz while the demodulator was set up for 48 kHz: MskDemodulator re-applies its settings
    with that rate (mskdemodulator.cpp:528-537), the adaptor replaces its bank; AeroL prints the same signal units."""
    pay = AF.random_payloads(8, 1200, seed=11)
    bits, _ = AF.p_channel_bits(pay, 1200)
    n = int(len(bits) * 24000 / 1200) + 2000
    pcm, _ = G.msk(n, fb=1200.0, Fs=24000.0, fc=1007.0, ebno_db=16.0, seed=33, bits=np.concatenate([bits, np.zeros(16, np.uint8)]))
    ref = run_demo("ref", "msk", pcm, fb=1200, datarate=24000)
    hip = run_demo("hip", "msk", pcm, fb=1200, datarate=24000)
    assert len(ref) > 500
    assert hip == ref
