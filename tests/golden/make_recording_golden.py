"""The reference's own sample recording of a 10.5 kbps P channel (samples/10.5k_sample.ogg, Ogg Vorbis, mono, 44.1 kHz) as a fixture:
the first seconds decoded (scripts/vorbis_decode.py -- this image has no audio decoder), resampled to the 48 kHz the demodulator works at
(scipy.signal.resample_poly 160 / 147: what a sound card does when the file is played into JAERO), and what the UNMODIFIED reference
(oracle/_ref: OqpskDemodulator, then AeroL) makes of exactly that PCM.

Run in the build container only (needs /root/reference, /opt/conda Qt, scipy):  python tests/golden/make_recording_golden.py
recording_oqpsk_10k5.npz: pcm int16 [n]; soft = the soft bits handed to processDemodulatedSoftBits; status = one row per
FreqOffsetEstimateSlot; sus = the signal units the reference's AeroL printed for those soft bits ([k, 10 bytes, crc ok] per row).
The decoder is not bit-exact with libvorbis and need not be: the PCM stored here is the common input of both sides.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from oracle import oracle as O  # noqa: E402

SECONDS = 12.0


def main():
    from scipy.signal import resample_poly
    import vorbis_decode

    assert O.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    x, rate = vorbis_decode.decode("/root/reference/samples/10.5k_sample.ogg", SECONDS * 1.02 + 0.5)
    assert rate == 44100 and x.shape[0] == 1
    y = resample_poly(x[0], 160, 147)
    pcm = np.clip(np.round(y * 32767.0), -32768, 32767).astype(np.int16)[: int(SECONDS * 48000)]
    ref = O.run_ref("oqpsk", pcm)
    sus, _txt = O.run_ref_aerol(10500, ref["soft"], 32)
    rows = np.array([[k] + list(b) + [int(ok)] for k, b, ok in sus], dtype=np.int16).reshape(-1, 12)
    np.savez_compressed(os.path.join(HERE, "recording_oqpsk_10k5.npz"), pcm=pcm, kind="oqpsk", opts=np.array(repr({})), soft=ref["soft"],
                        status=ref["status"], sus=rows)
    print("recording_oqpsk_10k5: pcm", pcm.shape, "peak", int(np.abs(pcm).max()), "soft", ref["soft"].shape, "status", ref["status"].shape,
          "signal units", rows.shape, "crc ok", int(rows[:, 11].sum()), "carrier %.1f Hz" % ref["status"][-1, 1])


if __name__ == "__main__":
    main()
