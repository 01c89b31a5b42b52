"""GPU (-m gpu): the three Aero-L banks at the size `bench.py --workload aerol | aerol_burst | aerol_c` times them -- 65 536 channels,
Viterbi layout chosen BY SIZE (the lane layout from 16 384 blocks on; nothing forced) -- against the oracle on the same soft bits:
signal-unit rows, R/T packets, voice rows and event rows exactly equal on channels spread over the bank (wave edges, neighbouring
wavefronts, the last one).  The banks of at most 70 channels in test_gpu_aerol*.py force the layout; these do not.

Every write is ragged: a channel's count depends on the stream it carries, so block ends, unique words and trial lengths fall anywhere
relative to the writes.  Reference: JAERO/aerol.cpp:1124-1600 (Decode), :2187-2502 (DecodeC), JAERO/aerol.h:631-879 (R/T packets)."""
import importlib.util
import os

import numpy as np
import pytest

from jaero_amd import aerol_frames as AF

pytestmark = pytest.mark.gpu
NCH = 65536
CHECK = sorted({0, 1, 63, 64, 255, 256, 257, 511, 1023, 4095, 4096, 16383, 16384, 16385, 30000, 32767, 32768, 50001, 65471, 65535})


@pytest.fixture(scope="module")
def B():
    from jaero_amd import capi
    from jaero_amd import demodulator as D

    capi.lib()  # fail loudly if the extension is missing
    return D


def feed_ragged(bank, streams, idx, chunk, rng):
    """streams: the distinct soft-bit streams; channel c carries streams[idx[c]].  Per write every STREAM draws its own count, every
    channel carrying it gets that many soft bits; rows are expanded on the device (the host never holds a [65536][chunk] buffer)."""
    import torch

    dev = torch.device("cuda", 0)
    nu = len(streams)
    lens = np.array([len(x) for x in streams])
    pos = np.zeros(nu, dtype=np.int64)
    idx_t = torch.from_numpy(np.asarray(idx, dtype=np.int64)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    while (pos < lens).any():
        cnt = np.minimum(rng.integers(chunk // 3, chunk + 1, size=nu), lens - pos).astype(np.int32)
        seg = np.zeros((nu, chunk), np.int16)
        for u in range(nu):
            seg[u, :cnt[u]] = streams[u][pos[u]:pos[u] + cnt[u]]
        rows = torch.from_numpy(seg).to(dev)[idx_t].contiguous()
        counts = torch.from_numpy(cnt).to(dev)[idx_t].contiguous()
        bank.write_device(rows.data_ptr(), counts.data_ptr(), chunk, chunk, stream)
        torch.cuda.synchronize()
        pos += cnt


def test_aerol_65536_channels(B, oracle_mod):
    """P channel, 10.5 kbps: 61 distinct streams (frames, noise levels, arm inversions, garbage prefixes, two noise-only), channel c carries
    stream (7 c) mod 61, so neighbouring lanes / wavefronts differ."""
    fb, nu = 10500, 61
    rng = np.random.default_rng(65536)
    streams = []
    for u in range(nu):
        if u % 30 == 29:
            streams.append(np.clip(np.round(128 + rng.normal(0, 40, 17000)), 0, 255).astype(np.int16))
            continue
        bits, _ = AF.p_channel_bits(AF.random_payloads(5, fb, seed=4000 + u), fb, invert_i=bool(u & 1), invert_q=bool(u & 2))
        pre = rng.integers(0, 2, size=int(rng.integers(0, 900)), dtype=np.uint8)
        streams.append(AF.to_soft(np.concatenate([pre, bits]), sigma=float(rng.uniform(0, 40)), seed=u))
    idx = (np.arange(NCH) * 7) % nu
    bank = B.AeroLBank(NCH, fb, max_softbits_per_write=6000, su_capacity=200)
    feed_ragged(bank, streams, idx, 6000, rng)
    nclean, refs = 0, {}
    for c in CHECK:
        u = int(idx[c])
        if u not in refs:
            refs[u] = oracle_mod.run_aerol(fb, streams[u], 1 << 20)
        o = refs[u]
        assert np.array_equal(bank.read_sus(c), o["sus"]), c
        assert np.array_equal(bank.read_events(c), o["events"]), c
        nclean += int(o["sus"][:, 14].sum())
    assert nclean > len(CHECK) * 20
    bank.close()


def test_aerol_c_65536_channels(B, oracle_mod):
    """C channel, 8400 bps (AeroL::DecodeC): 37 distinct streams at different frame phases, inversions and noise levels."""
    nu = 37
    rng = np.random.default_rng(8400)
    streams = [AF.c_channel_case(7000 + u, 3 + u % 2, 10.0 + 5.0 * (u % 7), inv=(bool(u & 1), bool(u & 2)), lead=int(rng.integers(0, 4200)))[1] for u in range(nu)]
    idx = (np.arange(NCH) * 5) % nu
    bank = B.AeroLBank(NCH, 8400, max_softbits_per_write=5000, su_capacity=40)
    feed_ragged(bank, streams, idx, 5000, rng)
    refs, nvoice = {}, 0
    for c in CHECK:
        u = int(idx[c])
        if u not in refs:
            a = oracle_mod.AeroL(8400)
            for s in range(0, len(streams[u]), 32):
                a.write(streams[u][s:s + 32])
            fn, voice = a.take_voice()
            refs[u] = (fn, voice, a.take_sus(), a.take_events())
        ofn, ovoice, osus, oev = refs[u]
        fn, voice = bank.read_voice(c)
        assert np.array_equal(fn, ofn) and np.array_equal(voice, ovoice), c
        assert np.array_equal(bank.read_sus(c), osus), c
        assert np.array_equal(bank.read_events(c), oev), c
        nvoice += len(ofn)
    assert nvoice >= len(CHECK)
    bank.close()


def test_aerol_burst_65536_channels(B, oracle_mod):
    """R/T packet search behind a burst demodulator, 10.5 kbps: 41 distinct burst streams (R and T packets of several lengths, lost tails,
    late unique words, three noise-only)."""
    spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    nu = 41
    rng = np.random.default_rng(41)
    streams = []
    for u in range(nu):
        if u % 14 == 13:
            streams.append(np.clip(np.round(128 + rng.normal(0, 40, 20000)), 0, 255).astype(np.int16))
        else:
            streams.append(mk.rt_case(300 + u, float(rng.uniform(8, 40)), (bool(u & 1), bool(u & 2)), cut=(u % 3 == 0))[1])
    idx = (np.arange(NCH) * 3) % nu
    bank = B.AeroLBank(NCH, 10500, max_softbits_per_write=3000, su_capacity=400, burst=True)
    feed_ragged(bank, streams, idx, 3000, rng)
    refs, npk = {}, 0
    for c in CHECK:
        u = int(idx[c])
        if u not in refs:
            o = oracle_mod.run_aerol_burst(10500, streams[u])
            refs[u] = (oracle_mod.packets_from_rows(o["packets"]), o["events"])
        want, oev = refs[u]
        assert bank.read_packets(c) == want, c
        assert np.array_equal(bank.read_events(c), oev), c
        npk += len(want)
    assert npk > len(CHECK)
    bank.close()
