"""GPU (-m gpu): parity at the bank sizes BASELINE.json names -- 4096-channel continuous OQPSK (configs[2]), 256-channel 1200 bps MSK
(configs[1]; plus 1024 channels so the persistent coarse kernel runs several estimates per workgroup), 4096-channel burst OQPSK
(configs[3]) -- and the two bundled recordings from their first to their last sample.

With more channels than the chip has CUs the coarse-frequency kernels (k_coarse6, k_coarse6_13) and k_trident run as persistent
workgroups that each take several estimates from the list and consume the ring / y[] rows they prefetched for the NEXT estimate; the
banks of at most 70 channels in the other test files never reach that code.  Channels are compared with their own oracle run
(O.run_demod / O.run_burst, exactly as test_gpu_parity.compare does) at indices spread over the bank: lanes 0 and 63 of a wavefront,
neighbouring wavefronts, workgroup iterations 1, 2, 3, ... of the persistent kernels, the last channel.  Every channel has its own
carrier, bits and noise and -- for the continuous kinds -- its own freq_center / lockingbw, so a mixed-up channel index cannot hide."""
import numpy as np
import pytest

from conftest import assert_soft_bytes, load_golden
from test_gpu_parity import compare

pytestmark = pytest.mark.gpu
SYM_TOL = 1e-5


@pytest.fixture(scope="module")
def B():
    from jaero_amd import capi
    from jaero_amd import demodulator as D

    capi.lib()  # fail loudly if the extension is missing
    return D


def spread(nch, extra=()):
    """Channel indices spread over a bank: wave edges, neighbouring waves, multiples of the CU count, the end."""
    want = [0, 1, 63, 64, 127, 255, 256, 257, 300, 511, 512, 767, 1023, 1024, 2047, 2048, 2500, 3071, 3333, 4094, 4095, *extra]
    return sorted({c for c in want if c < nch} | {nch - 1})


def feed_frames(bank, pcm, chunk):
    """pcm: torch int16 [nsamples, nch] on the device (frame-major), written `chunk` samples at a time."""
    from jaero_amd import capi

    for s in range(0, pcm.shape[0], chunk):
        bank.write(pcm[s:s + chunk], layout=capi.PCM_FRAME_MAJOR)


@pytest.mark.parametrize("capture,chunk", [(False, 4096), (True, 3000)])
def test_oqpsk_4096_channels(B, oracle_mod, capture, chunk):
    """BASELINE configs[2]: 4096-channel synthetic 48 kHz 10.5 kbps OQPSK.  62 000 samples = 15 coarse estimates per channel, each
    k_coarse4 launch takes 4096 estimates on 256 workgroups (16 persistent iterations each).  capture=False is the instantiation
    bench.py times (EbNo meters on, no symbol capture); capture=True adds the soft-symbol comparison and ragged 3000-sample writes
    (the estimate then falls in the middle of a write)."""
    import torch

    from jaero_amd import signalgen as G

    O = oracle_mod
    nch, nsamp = 4096, 62000
    dev = torch.device("cuda", 0)
    pcm, _, _ = G.oqpsk_torch(nch, nsamp, dev, ebno_db=10.0, seed=G.SEED_BASE + 4096)
    fcs = [8000.0 + 5.0 * ((c * 7) % 11 - 5) for c in range(nch)]
    lbw = [10500.0 - 500.0 * (c % 3) for c in range(nch)]
    setts = [B.OqpskSettings(freq_center=fcs[c], lockingbw=lbw[c]) for c in range(nch)]
    bank = B.DemodulatorBank(setts, ebno=True, status_log=True, capture_symbols=capture, max_write_samples=chunk,
                             softbit_capacity=int(nsamp * 10500 / 48000) + 64)
    feed_frames(bank, pcm, chunk)
    nsoft = 0
    for c in spread(nch):
        x = pcm[:, c].cpu().numpy()
        ref = O.run_demod(O.oqpsk_settings(freq_center=fcs[c], lockingbw=lbw[c]), x, chunk=chunk, capture_symbols=capture)
        assert ref["status"].shape[0] == 15
        sym = bank.read_symbols(c) if capture else None
        compare(bank.read_softbits(c), sym, bank.read_status_log(c), ref)
        nsoft += len(ref["soft"])
    assert nsoft > 20 * 5000  # the channels did lock: soft bits were compared, not just empty streams
    bank.close()


@pytest.mark.parametrize("fb", [10500.0, 8400.0])
def test_oqpsk_65536_channels(B, oracle_mod, fb):
    """The bank bench.py times: 65 536 channels (four front / back pairs per workgroup, 256 persistent coarse estimates per workgroup and
    launch), per-channel carriers, bits, noise and symbol-clock phases, at 10.5 kbps and at 8400 bps (prefilter + pair kernel with the
    halves taking turns).  28 672 samples = 7 estimates per channel; 20 channels spread over the bank against their oracle runs: soft
    bits, status rows (freq_est, freq_center, mse, EbNo)."""
    import torch

    from jaero_amd import capi
    from jaero_amd import signalgen as G

    free, _ = torch.cuda.mem_get_info(0)
    if free < 250 * (1 << 30):
        pytest.skip("needs ~250 GB of free HBM")
    O = oracle_mod
    nch, chunk, nsteps = 65536, 4096, 7
    dev = torch.device("cuda", 0)
    gen = G.OqpskTorchStream(nch, nsteps * chunk, dev, fb=fb, ebno_db=11.0, seed=G.SEED_BASE + 65536, nphase=32)
    st = B.OqpskSettings(fb=fb, lockingbw=fb, coarsefreqest_fft_power=14)
    bank = B.DemodulatorBank(st, nch, ebno=True, status_log=True, max_write_samples=chunk, softbit_capacity=int(nsteps * chunk * fb / 48000) + 64)
    check = sorted({0, 63, 64, 255, 256, 257, 511, 1023, 4095, 4096, 16383, 16384, 16385, 30000, 32767, 32768, 50001, 65471, 65472, 65535})
    cidx = torch.tensor(check, dtype=torch.long, device=dev)
    host = []
    for i in range(nsteps):
        blk = gen.render(i * chunk, chunk)
        host.append(blk[:, cidx].cpu().numpy())
        bank.write(blk, layout=capi.PCM_FRAME_MAJOR)
        del blk
    x = np.concatenate(host)  # [nsamples][len(check)]
    nsoft = 0
    for k, c in enumerate(check):
        ref = O.run_demod(O.oqpsk_settings(fb=fb, lockingbw=fb), np.ascontiguousarray(x[:, k]), chunk=chunk)
        assert ref["status"].shape[0] == nsteps
        compare(bank.read_softbits(c), None, bank.read_status_log(c), ref)
        nsoft += len(ref["soft"])
    assert nsoft > len(check) * 1000
    bank.close()


def test_msk_65536_channels(B, oracle_mod):
    """The bank `bench.py --workload msk` times: 65 536 channels of 1200 bps MSK (four k_msk_samples wavefronts per CU, k_coarse6_13 with
    256 persistent estimates per workgroup and launch, two launches per 4096-sample write).  37 distinct signals, channel c carries
    signal (5 c) mod 37 (so that neighbouring lanes, wavefronts and workgroup iterations all differ); 20 spread channels against the
    oracle run of their signal."""
    import torch

    from jaero_amd import capi
    from jaero_amd import signalgen as G

    free, _ = torch.cuda.mem_get_info(0)
    if free < 200 * (1 << 30):
        pytest.skip("needs ~200 GB of free HBM")
    O = oracle_mod
    nch, chunk, nsteps, nuniq = 65536, 4096, 7, 37
    dev = torch.device("cuda", 0)
    nsamp = nsteps * chunk
    uniq = np.stack([G.msk(nsamp, fb=1200.0, fc=1000.0 + 4.0 * (u % 9 - 4), ebno_db=11.0, seed=G.SEED_BASE + 650 + u)[0] for u in range(nuniq)])
    idx = (torch.arange(nch, device=dev) * 5) % nuniq
    pcm = torch.from_numpy(np.ascontiguousarray(uniq.T)).to(dev)[:, idx].contiguous()  # frame-major [nsamp, nch]
    bank = B.DemodulatorBank(B.MskSettings(fb=1200.0, lockingbw=1800.0, freq_center=1000.0), nch, ebno=True, status_log=True,
                             max_write_samples=chunk, softbit_capacity=int(nsamp * 1200 / 48000) + 64)
    for s0 in range(0, nsamp, chunk):
        bank.write(pcm[s0:s0 + chunk], layout=capi.PCM_FRAME_MAJOR)
    check = sorted({0, 63, 64, 255, 256, 257, 511, 1023, 4095, 4096, 16383, 16384, 16385, 30000, 32767, 32768, 50001, 65471, 65472, 65535})
    refs = {}
    nsoft = 0
    for c in check:
        u = (c * 5) % nuniq
        if u not in refs:
            refs[u] = O.run_demod(O.msk_settings(fb=1200.0, lockingbw=1800.0), uniq[u], chunk=chunk)
        compare(bank.read_softbits(c), None, bank.read_status_log(c), refs[u])
        nsoft += len(refs[u]["soft"])
    assert nsoft > len(check) * 300
    bank.close()


@pytest.mark.parametrize("nch,nsamp", [(256, 50000), (1024, 30000)])
def test_msk_1200_banks(B, oracle_mod, nch, nsamp):
    """BASELINE configs[1]: 256-channel synthetic 48 kHz 1200 bps MSK (one estimate per workgroup and launch), and 1024 channels
    (two persistent iterations of k_coarse6_13 per workgroup); an estimate every 2048 samples."""
    from jaero_amd import signalgen as G

    O = oracle_mod
    pcm, _, _ = G.channel_bank("msk", nch, nsamp, ebno_db=10.0, seed0=G.SEED_BASE + 256 + nch, fb=1200.0)
    fcs = [1000.0 + 3.0 * ((c * 5) % 9 - 4) for c in range(nch)]
    lbw = [1800.0 - 100.0 * (c % 3) for c in range(nch)]
    setts = [B.MskSettings(fb=1200.0, freq_center=fcs[c], lockingbw=lbw[c]) for c in range(nch)]
    chunk = 4096
    bank = B.DemodulatorBank(setts, ebno=True, status_log=True, capture_symbols=True, max_write_samples=chunk, softbit_capacity=nsamp)
    for s in range(0, nsamp, chunk):
        bank.write(pcm[:, s:s + chunk])
    for c in spread(nch):
        ref = O.run_demod(O.msk_settings(freq_center=fcs[c], lockingbw=lbw[c], fb=1200.0), pcm[c], chunk=chunk, capture_symbols=True)
        assert ref["status"].shape[0] == nsamp // 2048
        compare(bank.read_softbits(c), bank.read_symbols(c), bank.read_status_log(c), ref)
    bank.close()


def test_burst_oqpsk_4096_channels(B, oracle_mod):
    """BASELINE configs[3]: 4096-channel 10.5 kbps burst OQPSK, one burst per second and channel.  All bursts start within 600 symbols
    of each other, so the preamble ("trident") checks of the whole bank crowd into two or three segments: k_trident's persistent
    workgroups then take ~8 events each from the device-side list."""
    import torch

    from jaero_amd import signalgen as G
    from test_gpu_burst import check_events, check_soft

    O = oracle_mod
    nch, nsamp, chunk = 4096, 110000, 4096
    dev = torch.device("cuda", 0)
    pcm, _, _ = G.burst_oqpsk_torch(nch, nsamp, dev, ndata_sym=1500, ebno_db=15.0, seed=G.SEED_BASE + 40960, max_offset_sym=600)
    bank = B.DemodulatorBank(B.BurstOqpskSettings(), nch, capture_symbols=True, trace=True, max_write_samples=chunk, softbit_capacity=30000)
    feed_frames(bank, pcm, chunk)
    nacc = 0
    for c in spread(nch):
        x = pcm[:, c].cpu().numpy()
        ref = O.run_burst(O.burst_oqpsk_settings(), x, chunk=chunk, capture_symbols=True, trace=True)
        check_soft(bank.read_softbits(c), ref["soft"], f"channel {c}")
        check_events(bank.read_events(c), ref["events"])
        sym = bank.read_symbols(c)
        assert sym.shape == ref["symbols"].shape
        # The tracking chain reads the REAL part of the delayed analytic signal, and that part is one tap of the Hilbert kernel (-1 at
        # k = 1024): minus the input 1024 samples earlier.  A burst that opens its gate within the first L + D1 + D2 + 1024 = 12 537 samples
        # of a stream (L = 6145 the filter's latency, D1 + D2 = 5368 the two delay lines) is therefore demodulated, for up to 1024 samples =
        # 112 symbols, from input "before the stream began": exact zeros here (k_hilbert_fft copies the real part), while the reference's
        # FFT convolution leaves its round-off there (~1e-13 of full scale, whatever FFT library it is linked with), which the AGC at its
        # gain cap of 1.4e6 turns into "symbols" of 1e-7 .. 1e-4.  Those leading rows carry nothing but the reference's FFT round-off:
        # they must be exact zeros here, tiny there, at most 112, and nothing else may differ.
        lead = 0
        while lead < len(sym) and not sym[lead, :2].any():
            lead += 1
        assert lead <= 112 and np.abs(ref["symbols"][:lead, :2]).max(initial=0.0) < 2e-4
        assert np.max(np.abs(sym - ref["symbols"])[lead:], initial=0.0) < SYM_TOL
        nacc += int((ref["soft"] == -1).sum())
    assert nacc >= 5  # one whole burst per channel in view
    bank.close()


def test_burst_oqpsk_65536_channels(B, oracle_mod):
    """The bank `bench.py --workload burst_oqpsk` times: 65 536 channels, one burst per second and channel at random offsets
    (k_hilbert_fft with 8192 workgroups x 2 blocks per segment, k_trident's persistent workgroups with thousands of events per segment).
    20 spread channels against their oracle runs: soft bits (with the -1 markers) and events."""
    import torch

    from jaero_amd import signalgen as G
    from test_gpu_burst import check_events, check_soft

    free, _ = torch.cuda.mem_get_info(0)
    if free < 200 * (1 << 30):
        pytest.skip("needs ~200 GB of free HBM")
    O = oracle_mod
    nch, nsamp, chunk = 65536, 26 * 4096, 4096
    dev = torch.device("cuda", 0)
    pcm, _, _ = G.burst_oqpsk_torch(nch, nsamp, dev, ndata_sym=1500, ebno_db=15.0, seed=G.SEED_BASE + 65537)
    bank = B.DemodulatorBank(B.BurstOqpskSettings(), nch, max_write_samples=chunk, softbit_capacity=30000)
    feed_frames(bank, pcm, chunk)
    check = sorted({0, 63, 64, 255, 256, 257, 511, 1023, 4095, 4096, 16383, 16384, 16385, 30000, 32767, 32768, 50001, 65471, 65472, 65535})
    nacc = 0
    for c in check:
        x = pcm[:, c].cpu().numpy()
        ref = O.run_burst(O.burst_oqpsk_settings(), x, chunk=chunk)
        check_soft(bank.read_softbits(c), ref["soft"], f"channel {c}")
        check_events(bank.read_events(c), ref["events"])
        nacc += int((ref["soft"] == -1).sum())
    assert nacc >= 10
    bank.close()


# ---------------------------------------------------------------------------------------------------------------------------------
# the bundled recordings, whole (north_star: "bit-exact on the decoded differential bitstream for the bundled sample files")
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["1200bps_burst_sample1", "1200bps_burst_sample2"])
def test_recording_through_burst_msk(B, name):
    """samples/1200bps_burst_sample{1,2}.wav through a burst MSK bank = what the unmodified BurstMskDemodulator emitted
    (tests/golden/*_burstmsk.npz, made by oracle/_ref): every soft bit, every start-of-burst marker, every emission."""
    from test_gpu_burst import check_events, check_soft

    pcm = load_golden(name + "_pcm")["pcm"]
    g = load_golden(name + "_burstmsk")
    assert len(pcm) == int(g["nsamples"])
    chunk = 4096
    bank = B.DemodulatorBank(B.BurstMskSettings(freq_center=1000.0, fb=1200.0), 1, max_write_samples=chunk, softbit_capacity=60000)
    for s in range(0, len(pcm), chunk):
        bank.write(pcm[None, s:s + chunk])
    soft = bank.read_softbits(0)
    check_soft(soft, g["soft"])
    assert np.array_equal(soft, g["soft"])  # not just the hard decisions: these recordings reproduce byte for byte
    ev = bank.read_events(0)
    ev[:, 0] = np.floor(ev[:, 0] / chunk) * chunk  # the reference driver stamps emissions with their write's first sample
    check_events(ev, g["events"])
    assert int((soft == -1).sum()) >= 2  # the recordings hold several bursts each
    bank.close()


@pytest.mark.parametrize("name", ["1200bps_burst_sample1", "1200bps_burst_sample2"])
def test_recording_through_continuous_msk(B, name):
    """The same recordings through the continuous 1200 bps MSK demodulator (BASELINE configs[0] stand-in) = what the
    unmodified MskDemodulator emitted (tests/golden/*_contmsk.npz): ~250 coarse estimates, ~13 000 soft bits."""
    pcm = load_golden(name + "_pcm")["pcm"]
    g = load_golden(name + "_contmsk")
    assert len(pcm) == int(g["nsamples"])
    chunk = 4096
    bank = B.DemodulatorBank(B.MskSettings(fb=1200.0, lockingbw=1800.0, freq_center=1000.0), 1, ebno=True, status_log=True,
                             max_write_samples=chunk, softbit_capacity=len(pcm))
    for s in range(0, len(pcm), chunk):
        bank.write(pcm[None, s:s + chunk])
    soft, log = bank.read_softbits(0), bank.read_status_log(0)
    n = len(g["soft"])
    assert n <= len(soft) < n + 12
    assert np.array_equal(soft[:n] >= 128, g["soft"] >= 128)
    assert_soft_bytes(soft[:n], g["soft"])
    assert log.shape == g["status"].shape
    assert np.array_equal(log[:, [0, 5]], g["status"][:, [0, 5]])
    assert np.max(np.abs(log[:, 1:4] - g["status"][:, 1:4])) < 1e-6
    bank.close()


def test_recordings_side_by_side_in_one_bank(B):
    """Both recordings as channels 0 and 65 of one 66-channel burst MSK bank (the rest silence / the other recording delayed):
    the per-lane burst gates of neighbouring channels open at different times."""
    from test_gpu_burst import check_soft

    a = load_golden("1200bps_burst_sample1_pcm")["pcm"]
    b = load_golden("1200bps_burst_sample2_pcm")["pcm"]
    n = min(len(a), len(b))
    n -= n % 4096
    nch = 66
    pcm = np.zeros((nch, n), np.int16)
    pcm[0], pcm[65] = a[:n], b[:n]
    pcm[1, 30000:] = b[:n - 30000]
    pcm[64, 77777:] = a[:n - 77777]
    bank = B.DemodulatorBank(B.BurstMskSettings(freq_center=1000.0, fb=1200.0), nch, max_write_samples=4096, softbit_capacity=60000)
    for s in range(0, n, 4096):
        bank.write(pcm[:, s:s + 4096])
    ga, gb = load_golden("1200bps_burst_sample1_burstmsk")["soft"], load_golden("1200bps_burst_sample2_burstmsk")["soft"]
    sa, sb = bank.read_softbits(0), bank.read_softbits(65)
    # the files were cut to a common length: what was emitted is a prefix of the whole-file golden
    assert len(sa) > 0.9 * len(ga) and len(sb) > 0.9 * len(gb)
    check_soft(sa, ga[:len(sa)])
    check_soft(sb, gb[:len(sb)])
    assert len(bank.read_softbits(2)) == 0  # silence never opens the gate
    bank.close()


def test_10500_and_8400_banks_alive_together(B, oracle_mod):
    """Two OQPSK banks with different matched filters (RRC alpha 1.0 at 10500 bps, 0.6 at 8400 bps) live in one process and are fed
    alternately: each keeps its own taps (they used to share one process-global constant symbol, so the bank created last
    silently changed the other's filter)."""
    from jaero_amd import signalgen as G

    O = oracle_mod
    nsamp, chunk = 60000, 4096
    p105, _ = G.oqpsk(nsamp, fc=8030.0, ebno_db=11.0, seed=G.SEED_BASE + 1)
    p84, _ = G.oqpsk(nsamp, fb=8400.0, fc=7975.0, ebno_db=11.0, seed=G.SEED_BASE + 2)
    b105 = B.DemodulatorBank(B.OqpskSettings(), 1, ebno=True, status_log=True, capture_symbols=True, max_write_samples=chunk, softbit_capacity=nsamp)
    b84 = B.DemodulatorBank(B.OqpskSettings(fb=8400.0, lockingbw=8400.0), 1, ebno=True, status_log=True, capture_symbols=True,
                            max_write_samples=chunk, softbit_capacity=nsamp)
    bb = B.DemodulatorBank(B.BurstOqpskSettings(), 1, max_write_samples=chunk, softbit_capacity=nsamp)  # a third user of 55-tap OQPSK filters
    for s in range(0, nsamp, chunk):
        b105.write(p105[None, s:s + chunk])
        b84.write(p84[None, s:s + chunk])
        bb.write(p105[None, s:s + chunk])
    r105 = O.run_demod(O.oqpsk_settings(), p105, chunk=chunk, capture_symbols=True)
    r84 = O.run_demod(O.oqpsk_settings(fb=8400.0, lockingbw=8400.0), p84, chunk=chunk, capture_symbols=True)
    compare(b105.read_softbits(0), b105.read_symbols(0), b105.read_status_log(0), r105)
    compare(b84.read_softbits(0), b84.read_symbols(0), b84.read_status_log(0), r84)
    assert len(r105["soft"]) > 5000 and len(r84["soft"]) > 4000
    for b in (b105, b84, bb):
        b.close()


def test_msk_family_banks_alive_together(B, oracle_mod):
    """Every MSK-family kernel reads its bank's own half-sine taps (the burst MSK loop used to read a process-global constant table
    keyed by fb, which every continuous MSK create overwrote -- with 20 / 40 taps at 24 / 12 kHz).  A burst MSK 1200 bank (80 taps) and
    a burst MSK 600 bank (160 taps) stay alive and are fed while continuous MSK banks at (24 kHz, 1200: 40 taps), (24 kHz, 600: 80
    taps) and (12 kHz, 1200: 20 taps) are created and run between their writes; all five equal their oracle runs."""
    from jaero_amd import signalgen as G
    from test_gpu_burst import check_events, check_soft

    O = oracle_mod
    rng = np.random.default_rng(77)
    burst, bpcm, bopts = {}, {}, {}
    for fb in (1200, 600):
        n = int(48000 * 4 * (1200 / fb))
        bpcm[fb] = G.burst_msk(n, burst_starts=[int(n * 0.2)], fb=float(fb), fc=1900.0 + 40.0 * (fb == 600), ncw=130, ebno_db=18.0,
                               seed=G.SEED_BASE + 770 + fb)[0]
        bopts[fb] = dict(freq_center=1000.0, fb=float(fb), lockingbw=1.5 * fb)
        burst[fb] = B.DemodulatorBank(B.BurstMskSettings(**bopts[fb]), 1, capture_symbols=True, max_write_samples=4096, softbit_capacity=30000)
    cont = [(24000.0, 1200.0), (24000.0, 600.0), (12000.0, 1200.0)]
    cbank, cpcm = {}, {}
    pos = {1200: 0, 600: 0}

    def feed_bursts(frac):
        for fb in (1200, 600):
            end = int(len(bpcm[fb]) * frac)
            while pos[fb] < end:
                m = min(4096, end - pos[fb])
                burst[fb].write(bpcm[fb][None, pos[fb]:pos[fb] + m])
                pos[fb] += m

    feed_bursts(0.1)
    for k, (Fs, fb) in enumerate(cont):
        nsamp = int(Fs * 4)
        cpcm[(Fs, fb)] = G.msk(nsamp, fb=fb, Fs=Fs, fc=1000.0 + 6.0 * k, ebno_db=12.0, seed=G.SEED_BASE + 780 + k)[0]
        # created while the burst banks are alive and mid-stream: must not touch their filters
        cbank[(Fs, fb)] = B.DemodulatorBank(B.MskSettings(fb=fb, lockingbw=1.5 * fb, freq_center=1000.0, Fs=Fs), 1, ebno=True, status_log=True,
                                            capture_symbols=True, max_write_samples=3000, softbit_capacity=nsamp)
        x = cpcm[(Fs, fb)]
        for s in range(0, nsamp, 3000):
            cbank[(Fs, fb)].write(x[None, s:s + 3000])
        feed_bursts(0.1 + 0.3 * (k + 1))
    feed_bursts(1.0)
    nacc = 0
    for fb in (1200, 600):
        ref = O.run_burst(O.burst_msk_settings(**bopts[fb]), bpcm[fb], chunk=4096, capture_symbols=True)
        check_soft(burst[fb].read_softbits(0), ref["soft"], f"burst MSK {fb}")
        check_events(burst[fb].read_events(0), ref["events"])
        sym = burst[fb].read_symbols(0)
        assert sym.shape == ref["symbols"].shape and np.max(np.abs(sym - ref["symbols"]), initial=0.0) < SYM_TOL
        nacc += int((ref["soft"] == -1).sum())
        assert len(ref["soft"]) > 300
    assert nacc == 2
    for (Fs, fb), bank in cbank.items():
        ref = O.run_demod(O.msk_settings(lockingbw=1.5 * fb, fb=fb, Fs=Fs), cpcm[(Fs, fb)], chunk=3000, capture_symbols=True)
        compare(bank.read_softbits(0), bank.read_symbols(0), bank.read_status_log(0), ref)
    for b in list(burst.values()) + list(cbank.values()):
        b.close()


def test_burst_msk_65536_channels(B, oracle_mod):
    """The bank `bench.py --workload burst_msk` times: 65 536 channels of 1200 bps burst MSK (k_burst_msk_fb pairs in two residency rounds,
    per-lane ring positions, k_trident's persistent workgroups).  37 distinct streams with their bursts at different offsets, channel c
    carries stream (5 c) mod 37; 20 spread channels against the oracle run of their stream: soft bits with markers, events."""
    import torch

    from jaero_amd import capi
    from jaero_amd import signalgen as G
    from test_gpu_burst import check_events, check_soft

    free, _ = torch.cuda.mem_get_info(0)
    if free < 200 * (1 << 30):
        pytest.skip("needs ~200 GB of free HBM")
    O = oracle_mod
    nch, chunk, nsteps, nuniq = 65536, 4096, 30, 37
    nsamp = nsteps * chunk
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(G.SEED_BASE + 6553)
    uniq = np.stack([G.burst_msk(nsamp, burst_starts=list(range(int(rng.integers(2000, 60000)), nsamp - 1000, 72000)), ndata=700, fb=1200.0,
                                 fc=1000.0 + float(rng.uniform(-8, 8)), ebno_db=18.0, seed=G.SEED_BASE + 6600 + u)[0] for u in range(nuniq)])
    idx = (torch.arange(nch, device=dev) * 5) % nuniq
    pcm = torch.from_numpy(np.ascontiguousarray(uniq.T)).to(dev)[:, idx].contiguous()  # frame-major [nsamp, nch]
    opts = dict(freq_center=1000.0, fb=1200.0)
    bank = B.DemodulatorBank(B.BurstMskSettings(**opts), nch, max_write_samples=chunk, softbit_capacity=int(nsamp * 1200 / 48000) + 64)
    for s0 in range(0, nsamp, chunk):
        bank.write(pcm[s0:s0 + chunk], layout=capi.PCM_FRAME_MAJOR)
    check = sorted({0, 63, 64, 255, 256, 257, 511, 1023, 4095, 4096, 16383, 16384, 16385, 30000, 32767, 32768, 50001, 65471, 65472, 65535})
    refs, nacc = {}, 0
    for c in check:
        u = (c * 5) % nuniq
        if u not in refs:
            refs[u] = O.run_burst(O.burst_msk_settings(**opts), uniq[u], chunk=chunk)
        check_soft(bank.read_softbits(c), refs[u]["soft"], f"channel {c}")
        check_events(bank.read_events(c), refs[u]["events"])
        nacc += int((refs[u]["soft"] == -1).sum())
    assert nacc >= len(check)
    bank.close()


@pytest.mark.parametrize("nch", [4096, 4099])
def test_msk_600_bank(B, oracle_mod, nch):
    """600 bps MSK at 48 kHz (the 160-tap loop, k_msk_fb<160,72,...,2,52>: two front / back pairs per workgroup) in a bank: 4096 channels =
    persistent iterations of k_coarse6_13 per workgroup and launch, 32 estimates per channel (one every 2048 samples); every channel its own
    lockingbw; 29 distinct signals, channel c carries signal (3 c) mod 29.  4099 channels = 65 groups (the last with three live lanes): the last of the 33
    two-pair workgroups holds one live pair and one that only keeps the barrier count."""
    import torch

    from jaero_amd import capi
    from jaero_amd import signalgen as G

    O = oracle_mod
    chunk, nuniq = 4096, 29
    nsamp = 16 * chunk
    dev = torch.device("cuda", 0)
    uniq = np.stack([G.msk(nsamp, fb=600.0, fc=1000.0 + 3.0 * (u % 7 - 3), ebno_db=11.0, seed=G.SEED_BASE + 6000 + u)[0] for u in range(nuniq)])
    idx = (torch.arange(nch, device=dev) * 3) % nuniq
    pcm = torch.from_numpy(np.ascontiguousarray(uniq.T)).to(dev)[:, idx].contiguous()
    lbw = [900.0 - 50.0 * (c % 3) for c in range(nch)]
    setts = [B.MskSettings(fb=600.0, lockingbw=lbw[c], freq_center=1000.0) for c in range(nch)]
    bank = B.DemodulatorBank(setts, ebno=True, status_log=True, max_write_samples=chunk, softbit_capacity=int(nsamp * 600 / 48000) + 64)
    for s0 in range(0, nsamp, chunk):
        bank.write(pcm[s0:s0 + chunk], layout=capi.PCM_FRAME_MAJOR)
    nsoft = 0
    for c in spread(nch):
        u = (c * 3) % nuniq
        ref = O.run_demod(O.msk_settings(fb=600.0, lockingbw=lbw[c]), uniq[u], chunk=chunk)
        assert ref["status"].shape[0] >= 3
        compare(bank.read_softbits(c), None, bank.read_status_log(c), ref)
        nsoft += len(ref["soft"])
    assert nsoft > 20 * 400
    bank.close()


RAGGED_NCH = 33091  # 518 wave groups (the last one with 3 of 64 channels) -> 130 four-pair workgroups, the last with 2 of its 4 pairs
RAGGED_CHECK = sorted({0, 63, 64, 4095, 16383, 16384, 20000, 30001, 32767, 32768, 33023, 33024, 33087, 33088, 33089, 33090})


@pytest.mark.parametrize("fb", [10500.0, 8400.0])
def test_oqpsk_ragged_last_workgroup(B, oracle_mod, fb):
    """The four-pair sample kernels (chosen above 32 768 channels) with a bank that does not fill their last workgroup: 33 091 channels =
    518 groups, so workgroup 129 holds pairs for groups 516 and 517 and two pairs that only keep the barrier count
    (k_oqpsk_fb.h `grp >= g.ngroups`; twice as many barriers at 8400 bps, where the halves take turns), and the last group has 3 live
    lanes.  Four coarse estimates per channel; 16 channels incl. the whole last group against their oracle runs."""
    import torch

    from jaero_amd import capi
    from jaero_amd import signalgen as G

    O = oracle_mod
    nch, chunk, nsteps = RAGGED_NCH, 4096, 4
    dev = torch.device("cuda", 0)
    gen = G.OqpskTorchStream(nch, nsteps * chunk, dev, fb=fb, ebno_db=11.0, seed=G.SEED_BASE + 33091, nphase=32)
    st = B.OqpskSettings(fb=fb, lockingbw=fb, coarsefreqest_fft_power=14)
    bank = B.DemodulatorBank(st, nch, ebno=True, status_log=True, max_write_samples=chunk, softbit_capacity=int(nsteps * chunk * fb / 48000) + 64)
    cidx = torch.tensor(RAGGED_CHECK, dtype=torch.long, device=dev)
    host = []
    for i in range(nsteps):
        blk = gen.render(i * chunk, chunk)
        host.append(blk[:, cidx].cpu().numpy())
        bank.write(blk, layout=capi.PCM_FRAME_MAJOR)
        del blk
    x = np.concatenate(host)
    nsoft = 0
    for k, c in enumerate(RAGGED_CHECK):
        ref = O.run_demod(O.oqpsk_settings(fb=fb, lockingbw=fb), np.ascontiguousarray(x[:, k]), chunk=chunk)
        assert ref["status"].shape[0] == nsteps
        compare(bank.read_softbits(c), None, bank.read_status_log(c), ref)
        nsoft += len(ref["soft"])
    assert nsoft > len(RAGGED_CHECK) * 500
    bank.close()


def test_msk_ragged_last_workgroup(B, oracle_mod):
    """k_msk_fb<80,32,...,4,22> (1200 bps MSK above 32 768 channels) with the same ragged bank: the back halves of absent groups must
    keep the barrier count while the front halves of live ones wait for their partial filter sums."""
    import torch

    from jaero_amd import capi
    from jaero_amd import signalgen as G

    O = oracle_mod
    nch, chunk, nsteps, nuniq = RAGGED_NCH, 4096, 4, 37
    dev = torch.device("cuda", 0)
    nsamp = nsteps * chunk
    uniq = np.stack([G.msk(nsamp, fb=1200.0, fc=1000.0 + 4.0 * (u % 9 - 4), ebno_db=11.0, seed=G.SEED_BASE + 330 + u)[0] for u in range(nuniq)])
    idx = (torch.arange(nch, device=dev) * 5) % nuniq
    pcm = torch.from_numpy(np.ascontiguousarray(uniq.T)).to(dev)[:, idx].contiguous()
    bank = B.DemodulatorBank(B.MskSettings(fb=1200.0, lockingbw=1800.0, freq_center=1000.0), nch, ebno=True, status_log=True,
                             max_write_samples=chunk, softbit_capacity=int(nsamp * 1200 / 48000) + 64)
    for s0 in range(0, nsamp, chunk):
        bank.write(pcm[s0:s0 + chunk], layout=capi.PCM_FRAME_MAJOR)
    refs, nsoft = {}, 0
    for c in RAGGED_CHECK:
        u = (c * 5) % nuniq
        if u not in refs:
            refs[u] = O.run_demod(O.msk_settings(fb=1200.0, lockingbw=1800.0), uniq[u], chunk=chunk)
        assert refs[u]["status"].shape[0] == 2 * nsteps  # 2^13-point estimates every 2048 samples
        compare(bank.read_softbits(c), None, bank.read_status_log(c), refs[u])
        nsoft += len(refs[u]["soft"])
    assert nsoft > len(RAGGED_CHECK) * 100
    bank.close()
