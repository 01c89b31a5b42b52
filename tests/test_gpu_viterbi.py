"""GPU (-m gpu): the batched Viterbi kernels -- one block per wavefront (k_viterbi) and one block per lane (k_viterbi_lanes, what
banks of >= 16384 blocks use) -- vs the libcorrect restatement (integer: bit-exact).  The test hook jaero_debug_viterbi_layout forces a layout."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["wave", "lanes"])
def layout(request, force_viterbi_layout):
    force_viterbi_layout(request.param)
    return request.param


def _soft(coded, rng, sigma, amp=64):
    x = (coded.astype(float) * 2 - 1) + rng.normal(0, sigma, coded.shape)
    return np.clip(np.round(x * amp + 128), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("nsoft,nblk", [(5078, 9), (662, 133), (320, 5), (6080, 3), (64, 2), (4992, 70), (1152, 3)])
def test_decode_soft_blocks(oracle_mod, layout, nsoft, nblk):
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


def test_continuous_streams(oracle_mod, layout):
    """= JConvolutionalCodec::Decode_Continuous per stream, overlap state carried across calls; full-size 10.5k blocks
    (5078 soft bytes) and 1200 bps blocks (662)."""
    from jaero_amd import capi

    O, L = oracle_mod, capi.lib()
    for nsoft in (5078, 662, 4992):
        nstreams = 7
        rng = np.random.default_rng(nsoft + 1)
        ov = np.zeros((nstreams, 64), np.uint8)
        codecs = [O.Codec(24) for _ in range(nstreams)]
        for it in range(4):
            if it == 2:  # streams 4.. restart: blocks with and without an overlap prefix in the same launch
                ov[4:] = 0
                codecs[4:] = [O.Codec(24) for _ in range(nstreams - 4)]
            soft = rng.integers(0, 256, size=(nstreams, nsoft), dtype=np.uint8)
            out = np.zeros((nstreams, nsoft // 2), np.uint8)
            nb = np.zeros(nstreams, np.int32)
            capi.check(L.jaero_viterbi_continuous(0, soft.ctypes.data, nstreams, nsoft, 24, ov.ctypes.data, out.ctypes.data,
                                                  nb.ctypes.data, 0, None))
            for s in range(nstreams):
                ref = codecs[s].decode_continuous(soft[s])
                assert len(ref) == nb[s]
                k = nb[s] - 8 if (it == 0 or (it == 2 and s >= 4)) else nb[s]  # tail of a stream's first block is undefined in the reference
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


def test_lane_layout_bank_vs_wave_layout(oracle_mod, force_viterbi_layout):
    """20000 blocks (the size at which the library switches to one block per lane by itself; last wavefront ragged): every decoded bit
    equal to the one-block-per-wavefront kernel's, and a sample of blocks equal to the oracle's."""
    import torch

    from jaero_amd import capi

    O, L = oracle_mod, capi.lib()
    nblk, nsoft = 20000, 1210
    g = torch.Generator(device="cuda").manual_seed(7)
    soft = torch.randint(0, 256, (nblk, nsoft), device="cuda", dtype=torch.uint8, generator=g)
    soft[::3] = (soft[::3] // 64) * 64 + 31   # coarse values: many exact metric ties
    soft[5] = 128
    soft[6] = 255
    soft[7] = 0
    outs = {}
    for lay in ("wave", "auto"):
        force_viterbi_layout(lay)
        out = torch.zeros((nblk, nsoft // 2), dtype=torch.uint8, device="cuda")
        capi.check(L.jaero_viterbi_decode_soft(0, soft.data_ptr(), nblk, nsoft, out.data_ptr(), 1, None))
        torch.cuda.synchronize()
        outs[lay] = out
    assert bool((outs["wave"] == outs["auto"]).all())
    h = soft.cpu().numpy()
    o = outs["auto"].cpu().numpy()
    for b in (0, 3, 5, 6, 7, 63, 64, 19999):
        ref = O.Codec().decode_soft(h[b])
        assert np.array_equal(ref[: nsoft // 2 - 6], o[b, : nsoft // 2 - 6]), b
