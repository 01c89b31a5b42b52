"""GPU (-m gpu): batched wave-per-block Viterbi kernel vs the libcorrect restatement (integer: bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _soft(coded, rng, sigma, amp=64):
    x = (coded.astype(float) * 2 - 1) + rng.normal(0, sigma, coded.shape)
    return np.clip(np.round(x * amp + 128), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("nsoft,nblk", [(5078, 9), (662, 33), (320, 5), (6080, 3), (64, 2)])
def test_decode_soft_blocks(oracle_mod, nsoft, nblk):
    from jaero_amd import capi

    O, L = oracle_mod, capi.lib()
    rng = np.random.default_rng(nsoft)
    soft = np.zeros((nblk, nsoft), np.uint8)
    for b in range(nblk):
        msg = rng.integers(0, 256, max(1, (nsoft // 2 - 8) // 8), dtype=np.uint8)
        coded = O.encode_bits(msg)[:nsoft]
        soft[b, :] = 128
        soft[b, : len(coded)] = _soft(coded, rng, 0.3 + 0.1 * (b % 5))
    soft[0] = rng.integers(0, 256, nsoft)        # garbage block: exercises ties / renormalisation
    soft[-1] = 128                                # all erasures
    out = np.zeros((nblk, nsoft // 2), np.uint8)
    capi.check(L.jaero_viterbi_decode_soft(0, soft.ctypes.data, nblk, nsoft, out.ctypes.data, 0, None))
    for b in range(nblk):
        ref = O.Codec().decode_soft(soft[b])
        assert np.array_equal(ref[: nsoft // 2 - 6], out[b, : nsoft // 2 - 6]), b


def test_continuous_streams(oracle_mod):
    """= JConvolutionalCodec::Decode_Continuous per stream, overlap state carried across calls; full-size 10.5k blocks
    (5078 soft bytes) and 1200 bps blocks (662)."""
    from jaero_amd import capi

    O, L = oracle_mod, capi.lib()
    for nsoft in (5078, 662):
        nstreams = 7
        rng = np.random.default_rng(nsoft + 1)
        ov = np.zeros((nstreams, 64), np.uint8)
        codecs = [O.Codec(24) for _ in range(nstreams)]
        for it in range(4):
            soft = rng.integers(0, 256, size=(nstreams, nsoft), dtype=np.uint8)
            out = np.zeros((nstreams, nsoft // 2), np.uint8)
            nb = np.zeros(nstreams, np.int32)
            capi.check(L.jaero_viterbi_continuous(0, soft.ctypes.data, nstreams, nsoft, 24, ov.ctypes.data, out.ctypes.data,
                                                  nb.ctypes.data, 0, None))
            for s in range(nstreams):
                ref = codecs[s].decode_continuous(soft[s])
                assert len(ref) == nb[s]
                k = nb[s] - 8 if it == 0 else nb[s]  # tail of the very first block is undefined in the reference
                assert np.array_equal(ref[:k], out[s, :k])


def test_device_pointers_and_roundtrip_property(oracle_mod):
    """Device-resident input/output (torch tensors) and the size-independent property: decode(encode(m)) == m for
    4096 blocks at once."""
    import torch

    from jaero_amd import capi

    O, L = oracle_mod, capi.lib()
    rng = np.random.default_rng(5)
    nblk, nsoft = 4096, 5078
    msg = rng.integers(0, 256, (nsoft // 2 - 8) // 8, dtype=np.uint8)
    coded = O.encode_bits(msg)[:nsoft]
    base = np.full(nsoft, 128, np.uint8)
    base[: len(coded)] = np.where(coded > 0, 200, 56)
    soft = torch.from_numpy(np.tile(base, (nblk, 1))).cuda()
    noise = torch.randint(-40, 41, soft.shape, device="cuda", dtype=torch.int16)
    soft = (soft.to(torch.int16) + noise).clamp(0, 255).to(torch.uint8).contiguous()
    out = torch.zeros((nblk, nsoft // 2), dtype=torch.uint8, device="cuda")
    capi.check(L.jaero_viterbi_decode_soft(0, soft.data_ptr(), nblk, nsoft, out.data_ptr(), 1, None))
    torch.cuda.synchronize()
    want = torch.from_numpy(np.unpackbits(msg)).cuda()
    assert bool((out[:, : want.numel()] == want[None, :]).all())
