"""The reference's own off-air recording of a 10.5 kbps R/T (burst) channel (samples/10.5k_burst_sample.mp3, MPEG-1 Layer III, mono, 44.1 kHz,
33.7 s) as a fixture: the first seconds decoded (scripts/mp3_decode.py -- this image has no audio decoder), resampled to the 48 kHz the demodulator
works at (scipy.signal.resample_poly 160 / 147: what a sound card does when the file is played into JAERO), and what the UNMODIFIED reference
(oracle/_ref: BurstOqpskDemodulator, then AeroL in burst mode) makes of exactly that PCM.

Run in the build container only (needs /root/reference, /opt/conda Qt, scipy):  python tests/golden/make_burst_recording_golden.py
recording_burst_oqpsk_10k5.npz: pcm int16 [n]; soft = what BurstOqpskDemodulator handed to processDemodulatedSoftBits (start-of-burst markers -1
kept); events = its SignalStatus / EbNo / Plottables emissions; packets = the R / T packets the reference's AeroL printed for those soft bits, one
row per packet in the form of the other burst goldens ([type 1 = R / 2 = T, n, n printed, header / payload bytes ...], 317 wide); bad = the number of
" Bad R/T Packet" lines; decoder = the decoder's self-check counters.  The decoded PCM is the common input of all sides (a conforming MP3 decoder is
within one LSB of any other); that the decoder is right is shown by the reference itself: it finds CRC-clean T packets in it.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from oracle import oracle as O  # noqa: E402

SECONDS = 15.0
SOURCE = "/root/reference/samples/10.5k_burst_sample.mp3"


def decode_48k(seconds=None):
    """(int16 PCM at 48 kHz, the decoder's self-check counters)"""
    from scipy.signal import resample_poly
    import mp3_decode

    x, rate, info = mp3_decode.decode(SOURCE, None if seconds is None else seconds * 1.02 + 0.5)
    assert rate == 44100 and x.shape[0] == 1
    assert info["resyncs"] == 0 and info["reservoir_underruns"] == 0 and info["reservoir_overlaps"] == 0 and info["huffman_misfits"] == 0, info
    y = resample_poly(x[0], 160, 147)
    pcm = np.clip(np.round(y * 32767.0), -32768, 32767).astype(np.int16)
    return (pcm if seconds is None else pcm[: int(seconds * 48000)]), info


def packet_rows(pk):
    """the row form of the other burst goldens (make_golden.ref_rows): [type, n, n printed, 4 header bytes / 17 R bytes, 10 bytes per signal unit ...] padded to 317"""
    rows = []
    for p in pk:
        if p[0] == "R":
            rows.append([1, 17, 0] + list(p[1]) + [0] * (10 * 31 + 4 - 17))
        else:
            flat = [v for su in p[3] for v in su]
            rows.append([2, len(p[3]), p[2]] + list(p[1]) + flat + [0] * (10 * 31 - len(flat)))
    return np.array(rows, dtype=np.int32).reshape(-1, 317)


def main():
    assert O.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    pcm, info = decode_48k(SECONDS)
    ref = O.run_ref("burstoqpsk", pcm)
    pk, bad, _txt = O.run_ref_aerol_burst(10500, ref["soft"])
    rows = packet_rows(pk)
    np.savez_compressed(os.path.join(HERE, "recording_burst_oqpsk_10k5.npz"), pcm=pcm, kind="burstoqpsk", opts=np.array(repr({})), soft=ref["soft"],
                        events=ref["events"], packets=rows, bad=np.int32(bad), decoder=np.array(json.dumps(info)))
    print("recording_burst_oqpsk_10k5: pcm", pcm.shape, "peak", int(np.abs(pcm).max()), "soft", ref["soft"].shape, "bursts", int((ref["soft"] == -1).sum()),
          "events", ref["events"].shape, "packets (type, signal units)", [(int(r[0]), int(r[1])) for r in rows], "bad", bad, "decoder", info)


if __name__ == "__main__":
    main()
