"""CPU: the libcorrect restatement (oracle/viterbi_oracle.c).  PARITY UNPINNED (libcorrect is not in the reference
tree, no reference test touches it) -- what is pinned here: encode -> channel -> decode round trips, erasure
handling, the JConvolutionalCodec wrapper against the real jconvolutionalcodec.cpp (golden from oracle/_ref)."""
import numpy as np
import pytest

from conftest import load_golden


def _soft(coded, rng, sigma, amp=64):
    x = (coded.astype(float) * 2 - 1) + rng.normal(0, sigma, coded.shape)
    return np.clip(np.round(x * amp + 128), 0, 255).astype(np.uint8)


def test_encoder_known_answer(oracle_mod):
    """K=7 polys {109,79}: impulse response of the encoder = taps of the two polynomials (newest bit in LSB)."""
    O = oracle_mod
    coded = O.encode_bits(np.array([0x80], np.uint8))  # a single 1 followed by zeros
    # output pair t = (parity(sr&109), parity(sr&79)) with sr = 1<<t
    exp = []
    for t in range(7):
        sr = 1 << t
        exp += [bin(sr & 109).count("1") & 1, bin(sr & 79).count("1") & 1]
    assert list(coded[:14]) == exp
    assert not coded[14:].any()


@pytest.mark.parametrize("sigma", [0.0, 0.4, 0.7])
def test_roundtrip_block(oracle_mod, sigma):
    O = oracle_mod
    rng = np.random.default_rng(7)
    msg = rng.integers(0, 256, 300, dtype=np.uint8)
    coded = O.encode_bits(msg)
    bits = O.Codec().decode_soft(_soft(coded, rng, sigma))
    want = np.unpackbits(msg)
    # decode_soft returns nsoft/2 entries; the first 8*len are the message, the encoder's flush bits follow
    assert np.array_equal(bits[: want.size], want)


def test_erasures_and_hard_extremes(oracle_mod):
    O = oracle_mod
    rng = np.random.default_rng(8)
    msg = rng.integers(0, 256, 120, dtype=np.uint8)
    coded = O.encode_bits(msg)
    soft = np.where(coded > 0, 255, 0).astype(np.uint8)
    soft[::7] = 128  # 14 % erasures
    bits = O.Codec().decode_soft(soft)
    assert np.array_equal(bits[: 8 * 120], np.unpackbits(msg))
    # all-erasure block decodes to something finite (ties resolved deterministically)
    allz = O.Codec().decode_soft(np.full(512, 128, np.uint8))
    assert allz.shape[0] == 256 and set(np.unique(allz)) <= {0, 1}


def test_continuous_stream_equals_message(oracle_mod):
    """Decode_Continuous (jconvolutionalcodec.cpp:151-201) over consecutive 5078-soft-bit blocks: after the first
    block, block k returns coded-stream bits [k*2539-6, (k+1)*2539-6) (6-bit look-back)."""
    O = oracle_mod
    rng = np.random.default_rng(9)
    nblk, blen = 4, 5078
    msg = rng.integers(0, 256, nblk * blen // 16 + 8, dtype=np.uint8)
    coded = O.encode_bits(msg)[: nblk * blen]
    soft = _soft(coded, rng, 0.5)
    want = np.unpackbits(msg)
    codec = O.Codec(24)
    for k in range(nblk):
        out = codec.decode_continuous(soft[k * blen:(k + 1) * blen])
        if k == 0:
            assert len(out) == (blen + 24) // 2 - 25
            assert np.array_equal(out[: 2400], want[25: 25 + 2400])
        else:
            assert len(out) == blen // 2
            lo = k * (blen // 2) - 6
            assert np.array_equal(out[: blen // 2 - 40], want[lo: lo + blen // 2 - 40])


def test_wrapper_matches_reference_golden(oracle_mod):
    """Golden produced by the REAL JConvolutionalCodec (oracle/_ref viterbi_cont / viterbi_soft modes)."""
    O = oracle_mod
    g = load_golden("viterbi_cont")
    soft, blen, out = g["soft"], int(g["blocklen"]), g["out_cont"]
    codec = O.Codec(int(g["padding"]))
    p = 0
    for off in range(0, len(soft) - blen + 1, blen):
        n = int(np.frombuffer(out[p:p + 4].tobytes(), "<u4")[0])
        ref_bits = out[p + 4: p + 4 + n]
        p += 4 + n
        mine = codec.decode_continuous(soft[off:off + blen])
        assert len(mine) == n
        # the last bits of the first block come from bytes libcorrect never wrote (uninitialised in the reference)
        k = n - 8 if off == 0 else n
        assert np.array_equal(mine[:k], ref_bits[:k])
    blen2, out2 = int(g["blocklen_soft"]), g["out_soft"]
    p = 0
    for off in range(0, 3 * blen2, blen2):
        n = int(np.frombuffer(out2[p:p + 4].tobytes(), "<u4")[0])
        ref_bits = out2[p + 4: p + 4 + n]
        p += 4 + n
        mine = O.Codec().decode_soft(soft[off:off + blen2])
        assert len(mine) == n == blen2 // 2
        assert np.array_equal(mine[: n - 6], ref_bits[: n - 6])
