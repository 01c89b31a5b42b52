"""Regenerates the committed golden fixtures from the UNMODIFIED reference (oracle/_ref/jaero_ref).

Run in the build container only (needs /root/reference + /opt/conda Qt):  python tests/golden/make_golden.py
Each .npz holds the exact int16 PCM fed in, the driver options, and what the reference emitted
(soft bits passed to processDemodulatedSoftBits; one status row per FreqOffsetEstimateSlot).
fft_golden.npz holds the 16-point vectors of JAERO/tests/fftwrapper_tests.cpp:27-29 and
JAERO/tests/fftrwrapper_tests.cpp:28-30, parsed from those files (numbers only).
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from jaero_amd import signalgen as G  # noqa: E402
from oracle import oracle as O  # noqa: E402


def save(name, pcm, kind, opts, ref):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), pcm=pcm, kind=kind, opts=np.array(repr(opts)),
                        soft=ref["soft"], status=ref["status"])
    print(name, "pcm", pcm.shape, "soft", ref["soft"].shape, "status", ref["status"].shape)


def main():
    assert O.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    n = 72000  # 1.5 s
    pcm, _ = G.oqpsk(n, fc=8037.5, ebno_db=12.0, seed=G.SEED_BASE + 1)
    save("oqpsk_10k5_default", pcm, "oqpsk", {}, O.run_ref("oqpsk", pcm))
    opts = dict(afc=1, chunk=1000, dcd_at=40000)
    pcm, _ = G.oqpsk(n, fc=7961.0, ebno_db=9.0, seed=G.SEED_BASE + 2)
    save("oqpsk_10k5_afc_chunk1000_dcd", pcm, "oqpsk", opts, O.run_ref("oqpsk", pcm, **opts))
    pcm, _ = G.msk(n, fb=1200, fc=1012.0, ebno_db=14.0, seed=G.SEED_BASE + 3)
    save("msk_1200_default", pcm, "msk", dict(fb=1200, lockingbw=1800), O.run_ref("msk", pcm, fb=1200, lockingbw=1800))
    pcm, _ = G.msk(n, fb=600, fc=995.0, ebno_db=12.0, seed=G.SEED_BASE + 4)
    opts = dict(fb=600, lockingbw=900, chunk=777, dcd_at=30000)
    save("msk_600_chunk777_dcd", pcm, "msk", opts, O.run_ref("msk", pcm, **opts))
    opts = dict(cpureduce=1)
    pcm, _ = G.oqpsk(120000, fc=8020.0, ebno_db=12.0, seed=G.SEED_BASE + 5)
    save("oqpsk_10k5_cpureduce", pcm, "oqpsk", opts, O.run_ref("oqpsk", pcm, **opts))

    # 8400 bps C-channel branch of OqpskDemodulator (JFastFir prefilter, centre-weighted coarse window)
    opts = dict(fb=8400, lockingbw=8400)
    pcm, _ = G.oqpsk(150000, fb=8400.0, fc=8021.0, ebno_db=11.0, seed=G.SEED_BASE + 6)
    save("oqpsk_8400_default", pcm, "oqpsk", opts, O.run_ref("oqpsk", pcm, **opts))
    opts = dict(fb=8400, lockingbw=8400, afc=1, chunk=1500, dcd_at=90000)
    pcm, _ = G.oqpsk(150000, fb=8400.0, fc=7968.0, ebno_db=9.0, seed=G.SEED_BASE + 7)
    save("oqpsk_8400_afc_chunk1500_dcd", pcm, "oqpsk", opts, O.run_ref("oqpsk", pcm, **opts))

    # the reference's bundled recordings through the continuous MSK demodulator: outputs only (inputs stay in
    # /root/reference/samples; the test that uses them skips when that tree is absent)
    import wave
    for f in ("1200bps_burst_sample1.wav", "1200bps_burst_sample2.wav"):
        w = wave.open("/root/reference/samples/" + f)
        x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
        r = O.run_ref("msk", x, fb=1200, lockingbw=1800)
        np.savez_compressed(os.path.join(HERE, f.replace(".wav", "") + "_contmsk.npz"), nsamples=len(x),
                            crc=np.uint32(np.bitwise_xor.reduce(x.astype(np.uint16).astype(np.uint32) * np.arange(1, len(x) + 1, dtype=np.uint32))),
                            soft=r["soft"], status=r["status"])
        print(f, len(x), r["soft"].shape)

    # Viterbi through the reference's JConvolutionalCodec wrapper (libcorrect = oracle restatement: parity unpinned)
    rng = np.random.default_rng(99)
    msg = rng.integers(0, 256, size=3 * 317, dtype=np.uint8)
    coded = O.encode_bits(msg)
    nblk, blen = 3, 5078
    coded = coded[: nblk * blen]
    x = (coded.astype(float) * 2 - 1) + rng.normal(0, 0.6, coded.shape)
    soft = np.clip(np.round(x * 50 + 128), 0, 255).astype(np.uint8)
    out = O.ref_tool("viterbi_cont", soft, np.uint8, blocklen=blen, padding=24)
    blen2 = 5072  # Decode_soft writes whole bytes: keep size/2 a multiple of 8 (jconvolutionalcodec.cpp:107-117)
    out2 = O.ref_tool("viterbi_soft", soft[: 3 * blen2], np.uint8, blocklen=blen2)
    np.savez_compressed(os.path.join(HERE, "viterbi_cont.npz"), soft=soft, blocklen=blen, padding=24, out_cont=out,
                        blocklen_soft=blen2, out_soft=out2, msg=msg)
    print("viterbi", soft.shape, out.shape, out2.shape)

    # FFT golden vectors from the reference's own unit tests
    def parse(path, name):
        txt = open(path).read()
        m = re.search(name + r"\s*=\s*\{(.*?)\};", txt, re.S)
        body = m.group(1)
        if "cpx_type" in body:
            nums = re.findall(r"cpx_type\(([-0-9.e]+),([-0-9.e]+)\)", body)
            return np.array([complex(float(a), float(b)) for a, b in nums])
        return np.array([float(v) for v in body.split(",")])
    t1 = "/root/reference/JAERO/tests/fftwrapper_tests.cpp"
    t2 = "/root/reference/JAERO/tests/fftrwrapper_tests.cpp"
    np.savez(os.path.join(HERE, "fft_golden.npz"),
             c_input=parse(t1, "input"), c_forward=parse(t1, "expected_forward"), c_fb=parse(t1, "expected_forward_backwards"),
             r_input=parse(t2, "input"), r_forward=parse(t2, "expected_forward"), r_fb=parse(t2, "expected_forward_backwards"))
    print("fft golden ok")


def burst():
    """Burst demodulators: what the UNMODIFIED BurstOqpskDemodulator / BurstMskDemodulator emit.
    events rows = [sample index of the write that carried the emission, kind (0 SignalStatus, 1 EbNo, 2 Plottables), value]."""
    assert O.have_ref()
    n = 150000
    pcm, bursts = G.burst_oqpsk(n, burst_starts=[40000, 100000], ndata_sym=1000, fc=8037.5, ebno_db=15.0, seed=G.SEED_BASE + 40)
    for name, opts in (("burst_oqpsk_10k5_default", dict(chunk=4096)), ("burst_oqpsk_10k5_chunk1500", dict(chunk=1500))):
        r = O.run_ref("burstoqpsk", pcm, **opts)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), pcm=pcm, kind="burstoqpsk", opts=np.array(repr(opts)), soft=r["soft"],
                            events=r["events"], tx_bits=np.concatenate([b for _, b in bursts]), tx_starts=np.array([s for s, _ in bursts]))
        print(name, "soft", r["soft"].shape, "events", r["events"].shape)
    # an excerpt (two bursts) of the reference's bundled 1200 bps R/T-channel recording: the real signal the burst MSK
    # demodulator was written for; north_star: "bit-exact ... for the bundled sample files"
    import wave
    w = wave.open("/root/reference/samples/1200bps_burst_sample1.wav")
    x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)[140000:340000].copy()
    opts = dict(fb=1200, freq_center=1000, chunk=4096)
    r = O.run_ref("burstmsk", x, **opts)
    np.savez_compressed(os.path.join(HERE, "burst_msk_1200_sample1_excerpt.npz"), pcm=x, kind="burstmsk", opts=np.array(repr(opts)),
                        soft=r["soft"], events=r["events"], source=np.array("samples/1200bps_burst_sample1.wav[140000:340000]"))
    print("burst_msk_1200_sample1_excerpt", x.shape, r["soft"].shape, r["events"].shape)
    # whole files: outputs only (the inputs stay in /root/reference/samples)
    for f in ("1200bps_burst_sample1.wav", "1200bps_burst_sample2.wav"):
        w = wave.open("/root/reference/samples/" + f)
        x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
        r = O.run_ref("burstmsk", x, fb=1200, freq_center=1000, chunk=4096)
        np.savez_compressed(os.path.join(HERE, f.replace(".wav", "") + "_burstmsk.npz"), nsamples=len(x), soft=r["soft"], events=r["events"])
        print(f, len(x), r["soft"].shape, r["events"].shape)


def aerol():
    """Aero-L bit pipeline: what the UNMODIFIED AeroL (JAERO/aerol.cpp) makes of generated P-channel frames.
    sus rows = [k, 10 payload bytes, crc_ok] in the order AeroL::Decode printed them."""
    from jaero_amd import aerol_frames as AF
    assert O.have_ref()
    for fb, grp, inv in ((10500, 32, (True, False)), (1200, 12, (False, False)), (600, 12, (False, False))):
        pay = AF.random_payloads(7, fb, seed=fb)
        bits, _ = AF.p_channel_bits(pay, fb, invert_i=inv[0], invert_q=inv[1])
        soft = AF.to_soft(bits, sigma=22.0, seed=fb + 1)
        sus, _ = O.run_ref_aerol(fb, soft, grp)
        rows = np.array([[k] + list(b) + [int(ok)] for k, b, ok in sus], dtype=np.int32)
        np.savez_compressed(os.path.join(HERE, f"aerol_{fb}.npz"), soft=soft, fb=fb, group=grp, sus=rows,
                            payloads=np.array([[list(p) for p in fr] for fr in pay], dtype=np.uint8))
        print("aerol", fb, soft.shape, rows.shape, "crc ok", int(rows[:, 11].sum()))


def aerol_c():
    """Aero-L C channel (second half): what the UNMODIFIED AeroL::DecodeC hands to Voicesignal(data, hex) per frame
    and the signal units it prints.  The first frame after start-up is excluded from the voice fixture's last byte: its three low
    bits come from bytes of JConvolutionalCodec::decoded that libcorrect never wrote (jconvolutionalcodec.cpp:165-169, first call)."""
    assert O.have_ref()
    from jaero_amd import aerol_frames as AF
    frames, soft = AF.c_channel_case(8401, 7, 24.0, inv=(True, False))
    voice, sus, dcd, _ = O.run_ref_aerol_c(soft, 32)
    np.savez_compressed(os.path.join(HERE, "aerol_c_8400_a.npz"), soft=soft, group=32, voice=voice,
                        sus=np.array([list(b) for b in sus], dtype=np.uint8).reshape(-1, 10),
                        voice_in=np.stack([f[0] for f in frames]),
                        sus_in=np.array([[list(p) for p in f[1]] for f in frames], dtype=np.uint8))
    print("aerol_c", soft.shape, voice.shape, len(sus), dcd)


def rt_case(seed, sigma, inv=(False, False), cut=False):
    """A burst-demodulator soft-bit stream with R and T packets (see aerol_frames.rt_burst_stream)."""
    from jaero_amd import aerol_frames as AF
    rng = np.random.default_rng(seed)
    rb = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))
    pk = [("R", rb(17)), ("T", (rb(4), [rb(10) for _ in range(2)])), ("T", (rb(4), [rb(10) for _ in range(7)])), ("R", rb(17)),
          ("T", (rb(4), [rb(10) for _ in range(31)]))]
    x = AF.rt_burst_stream(pk, sigma=sigma, seed=seed, invert_i=inv[0], invert_q=inv[1], gap=11000 + 37 * seed)
    if cut:  # a burst whose tail is lost (the T packet never passes its CRCs) and one whose unique word comes too late after the marker
        k = int(np.where(x < 0)[0][2])
        x = np.concatenate([x[: k + 500], x[k + 2600:]])
        k = int(np.where(x < 0)[0][3])
        x = np.concatenate([x[: k + 1], np.full(300, 128, np.int16), x[k + 1:]])
    return pk, x


def ref_packets_as_rows(ref):
    """reference text -> comparable tuples: ('R', 17 bytes) / ('T', 4 header bytes, n, tuple of 10-byte SUs)"""
    return [tuple(p) if p[0] == "R" else (p[0], p[1], p[2], tuple(p[3])) for p in ref]


def rt_case_msk(seed, sigma, invert=False, cut=False):
    """600 / 1200 bps burst-demodulator stream with R packets and T packets of 4, 9 and 15 units + one unit too many for a count."""
    from jaero_amd import aerol_frames as AF
    rng = np.random.default_rng(seed)
    rb = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))
    pk = [("R", rb(17)), ("T", (rb(4), [rb(10) for _ in range(4)])), ("T", (rb(4), [rb(10) for _ in range(9)])), ("R", rb(17)),
          ("T", (rb(4), [rb(10) for _ in range(16)]))]
    x = AF.rt_burst_stream_msk(pk, sigma=sigma, seed=seed, invert=invert, gap=4000 + 31 * seed)
    if cut:  # a T packet that loses its tail and an R packet whose unique word comes more than 250 soft bits after the marker
        k = int(np.where(x < 0)[0][2])
        x = np.concatenate([x[: k + 700], x[k + 1500:]])
        k = int(np.where(x < 0)[0][3])
        x = np.concatenate([x[: k + 1], np.full(300, 128, np.int16), x[k + 1:]])
    return pk, x


def ref_rows(ref):
    rows = []
    for p in ref:
        if p[0] == "R":
            rows.append([1, 17, 0] + list(p[1]) + [0] * (10 * 31 + 4 - 17))
        else:
            flat = [v for su in p[3] for v in su]
            rows.append([2, len(p[3]), p[2]] + list(p[1]) + flat + [0] * (10 * 31 - len(flat)))
    return np.array(rows, dtype=np.int32).reshape(-1, 317)


def aerol_burst():
    """Row f2: what the UNMODIFIED AeroL in burst mode (R/T channel packets, 10500 bps) makes of generated bursts, fed in the groups a
    burst demodulator emits."""
    assert O.have_ref()
    for name, (seed, sigma, inv, cut) in {"a": (1, 18.0, (False, False), False), "b": (2, 30.0, (True, False), True)}.items():
        pk, x = rt_case(seed, sigma, inv, cut)
        ref, bad, txt = O.run_ref_aerol_burst(10500, x)
        dcd = [(int(a), int(b)) for a, b in __import__("re").findall(r"#DCD (\d) (\d+)", txt)]
        rows = []
        for p in ref:
            if p[0] == "R":
                rows.append([1, 17, 0] + list(p[1]) + [0] * (10 * 31 + 4 - 17))
            else:
                flat = [v for su in p[3] for v in su]
                rows.append([2, len(p[3]), p[2]] + list(p[1]) + flat + [0] * (10 * 31 - len(flat)))
        np.savez_compressed(os.path.join(HERE, f"aerol_burst_10500_{name}.npz"), soft=x, packets=np.array(rows, dtype=np.int32), bad=bad,
                            dcd=np.array(dcd, dtype=np.int64))
        print("aerol burst", name, x.shape, [(r[0], r[1]) for r in rows], "bad", bad, "dcd", len(dcd))
    for fb, seed, inv in ((1200, 3, False), (600, 4, True)):
        pk, x = rt_case_msk(seed, 22.0, invert=inv, cut=(fb == 600))
        ref, bad, txt = O.run_ref_aerol_burst(fb, x)
        dcd = [(int(a), int(b)) for a, b in __import__("re").findall(r"#DCD (\d) (\d+)", txt)]
        rows = ref_rows(ref)
        np.savez_compressed(os.path.join(HERE, f"aerol_burst_{fb}_a.npz"), soft=x, packets=rows, bad=bad, dcd=np.array(dcd, dtype=np.int64))
        print("aerol burst", fb, x.shape, [(int(r[0]), int(r[1])) for r in rows], "bad", bad, "dcd", len(dcd))



def jfastfir():
    """The reference's own JFastFir test vectors (JAERO/tests/jfastfir_data_{input,expected_output}.cpp: the input and what JAERO
    v1.0.4.11 produced for it, checked by JAERO/tests/jfastfir_tests.cpp:31-58 from sample 4096 on at 1e-5) as a binary fixture, after
    confirming that the JFFT stand-in behind oracle/_ref reproduces them."""
    import re

    def parse(path):
        txt = open(path).read()
        v = re.findall(r"cpx_type\(([-+0-9.eE]+),([-+0-9.eE]+)\)", txt)
        return np.array([complex(float(a), float(b)) for a, b in v], dtype=np.complex128)

    x = parse("/root/reference/JAERO/tests/jfastfir_data_input.cpp")
    want = parse("/root/reference/JAERO/tests/jfastfir_data_expected_output.cpp")
    assert len(x) == len(want) and len(x) > 4096
    got = O.ref_tool("fastfir", x, np.complex128, alpha=0.6, K=2048, nfft=4096, Fs=48000, fsym=5250)
    err = float(np.max(np.abs(got[4096:] - want[4096:])))
    assert err < 1e-5, err
    np.savez_compressed(os.path.join(HERE, "jfastfir.npz"), input=x, expected_output=want,
                        source=np.array("JAERO/tests/jfastfir_data_input.cpp, jfastfir_data_expected_output.cpp (v1.0.4.11 of JAERO)"))
    print("jfastfir", len(x), "shim max |err| from 4096 on:", err)


def recordings():
    """Inputs of the two bundled recordings (samples/1200bps_burst_sample{1,2}.wav, mono int16 @ 48 kHz) as test fixtures, so that the
    GPU box (which has no /root/reference) can run the WHOLE files through the burst-MSK and continuous-MSK banks against the
    `_burstmsk` / `_contmsk` outputs above.  Data vectors, not reference source; the crc stored in the `_contmsk` golden is checked."""
    import wave
    for f in ("1200bps_burst_sample1", "1200bps_burst_sample2"):
        w = wave.open("/root/reference/samples/" + f + ".wav")
        assert w.getnchannels() == 1 and w.getframerate() == 48000 and w.getsampwidth() == 2
        x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
        g = np.load(os.path.join(HERE, f + "_contmsk.npz"))
        crc = np.uint32(np.bitwise_xor.reduce(x.astype(np.uint16).astype(np.uint32) * np.arange(1, len(x) + 1, dtype=np.uint32)))
        assert int(g["nsamples"]) == len(x) and int(g["crc"]) == int(crc)
        np.savez_compressed(os.path.join(HERE, f + "_pcm.npz"), pcm=x, source=np.array("samples/" + f + ".wav"))
        print(f, len(x), os.path.getsize(os.path.join(HERE, f + "_pcm.npz")))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "aerolburst":
        aerol_burst()
    elif len(sys.argv) > 1 and sys.argv[1] == "burst":
        burst()
    elif len(sys.argv) > 1 and sys.argv[1] == "aerol":
        aerol()
    elif len(sys.argv) > 1 and sys.argv[1] == "aerolc":
        aerol_c()
    elif len(sys.argv) > 1 and sys.argv[1] == "recordings":
        recordings()
    elif len(sys.argv) > 1 and sys.argv[1] == "jfastfir":
        jfastfir()
    else:
        main()
        burst()
        aerol()
        aerol_burst()
        aerol_c()
        recordings()
        jfastfir()
