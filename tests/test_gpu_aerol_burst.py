"""GPU (-m gpu): AeroL in burst mode -- the R / T channel packet search (SURVEY 8 row f2, 10500 bps) -- through the C ABI against the
unmodified AeroL's goldens and the oracle: packet bytes, ' Bad R/T Packet' notices, DataCarrierDetect edges.  Integer work: exact."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import load_golden
from jaero_amd import aerol_frames as AF
from test_aerol_oracle import burst_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    from jaero_amd import capi
    from jaero_amd import demodulator as D

    capi.lib()
    return D


def feed(bank, streams, chunk, rng=None):
    nch = len(streams)
    n = max(len(x) for x in streams)
    pos = np.zeros(nch, dtype=np.int64)
    lens = np.array([len(x) for x in streams])
    while (pos < lens).any():
        cnt = np.minimum(chunk if rng is None else rng.integers(1, chunk + 1, size=nch), lens - pos).astype(np.int32)
        buf = np.zeros((nch, chunk), np.int16)
        for c in range(nch):
            buf[c, :cnt[c]] = streams[c][pos[c]:pos[c] + cnt[c]]
        bank.write(buf, cnt)
        pos += cnt


@pytest.mark.parametrize("name", ["10500_a", "10500_b", "1200_a", "600_a"])
def test_against_reference_golden(B, oracle_mod, name):
    g = load_golden(f"aerol_burst_{name}")
    fb = int(name.split("_")[0])
    bank = B.AeroLBank(1, fb, max_softbits_per_write=4000, su_capacity=600, burst=True)
    feed(bank, [g["soft"]], 4000)
    assert np.array_equal(burst_rows(bank.read_packets(0), msk=fb != 10500), g["packets"])
    ev = bank.read_events(0)
    assert int((ev[:, 1] == 3).sum()) == int(g["bad"])
    starts = np.array([s for s, _ in oracle_mod.demod_groups(g["soft"])])
    dcd = [(int(v), int(starts[np.searchsorted(starts, i, side="right") - 1])) for i, k, v in ev[1:] if k == 0]
    assert dcd == [tuple(r) for r in g["dcd"].tolist()]
    bank.close()


@pytest.mark.parametrize("layout", ["wave", "lanes"])
def test_bank_vs_oracle(B, oracle_mod, force_viterbi_layout, layout):
    """70 channels (two wave groups): different packets, noise levels, arm inversions, lost tails / late unique words, ragged write
    sizes (so trial lengths, markers and group ends fall anywhere relative to the writes).  Both Viterbi layouts: banks from 16 384
    channels on decode one trial per lane, the lanes of a wavefront grouped by trial length (k_viterbi_lanes, lens)."""
    force_viterbi_layout(layout)
    spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    nch = 70
    rng = np.random.default_rng(3)
    streams = []
    for c in range(nch):
        if c % 7 == 6:
            streams.append(np.clip(np.round(128 + rng.normal(0, 40, 30000)), 0, 255).astype(np.int16))  # noise only
        else:
            _, x = mk.rt_case(100 + c, float(rng.uniform(8, 45)), (bool(c & 1), bool(c & 2)), cut=(c % 3 == 0))
            streams.append(x)
    # rows of 3000 entries are 16-byte aligned (k_aerolb_bits takes the soft entries eight at a time), rows of 2996 are not (one at a time)
    width = 3000 if layout == "wave" else 2996
    bank = B.AeroLBank(nch, 10500, max_softbits_per_write=width, su_capacity=700, burst=True)
    feed(bank, streams, width, rng)
    npk = 0
    for c in range(nch):
        o = oracle_mod.run_aerol_burst(10500, streams[c])
        want = oracle_mod.packets_from_rows(o["packets"])
        assert bank.read_packets(c) == want, c
        assert np.array_equal(bank.read_events(c), o["events"]), c
        npk += len(want)
    assert npk > 150
    bank.close()


@pytest.mark.parametrize("width", [4096, 4100])
def test_bank_short_gaps(B, oracle_mod, width):
    """Bursts that follow each other while the frame countdown of the one before still runs: the next start-of-burst marker arrives in the
    stretch k_aerolb_bits<true> takes eight entries at a time (rows of 4096 entries; rows of 4100 are not 16-byte aligned and go bit by
    bit).  Markers at every position of an aligned group, ragged writes."""
    from test_aerolb_emul import short_gap_streams

    streams = short_gap_streams(16)
    rng = np.random.default_rng(12)
    bank = B.AeroLBank(len(streams), 10500, max_softbits_per_write=width, su_capacity=700, burst=True)
    feed(bank, streams, width, rng)
    npk = 0
    for c in range(len(streams)):
        o = oracle_mod.run_aerol_burst(10500, streams[c])
        want = oracle_mod.packets_from_rows(o["packets"])
        assert bank.read_packets(c) == want, c
        assert np.array_equal(bank.read_events(c), o["events"]), c
        npk += len(want)
    assert npk >= 2 * len(streams)
    bank.close()


@pytest.mark.parametrize("layout", ["wave", "lanes"])
@pytest.mark.parametrize("fb", [1200, 600])
def test_msk_bank_vs_oracle(B, oracle_mod, force_viterbi_layout, fb, layout):
    """600 / 1200 bps bursts (updateMSK: R test at 5 blocks, count peek at 11, decode at the announced length): 66 channels, ragged
    writes, inverted streams, lost tails, late unique words, noise-only channels."""
    spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    force_viterbi_layout(layout)
    nch = 66
    rng = np.random.default_rng(fb)
    streams = []
    for c in range(nch):
        if c % 11 == 10:
            streams.append(np.clip(np.round(128 + rng.normal(0, 40, 20000)), 0, 255).astype(np.int16))
        else:
            _, x = mk.rt_case_msk(200 + c, float(rng.uniform(8, 40)), invert=bool(c & 1), cut=(c % 3 == 0))
            streams.append(x)
    width = 2500 if layout == "wave" else 2504  # single loads / groups of eight in k_aerolb_bits
    bank = B.AeroLBank(nch, fb, max_softbits_per_write=width, su_capacity=700, burst=True)
    feed(bank, streams, width, rng)
    npk = 0
    for c in range(nch):
        o = oracle_mod.run_aerol_burst(fb, streams[c])
        want = oracle_mod.packets_from_rows(o["packets"])
        assert bank.read_packets(c) == want, c
        assert np.array_equal(bank.read_events(c), o["events"]), c
        npk += len(want)
    assert npk > 150
    bank.close()


def test_pcm_to_packets_on_device(B, oracle_mod):
    """R and T packets -> burst OQPSK passband PCM (carrier, alternating preamble, unique word, packet) -> burst demodulator bank ->
    the emitted soft bits stay in HBM -> burst-mode Aero-L bank.  The packets that come out equal what the oracle chain (demodulator
    restatement -> AeroL restatement) produces from the same PCM, and each of them is one of the transmitted ones, in order.  (Not every
    burst survives the reference's burst demodulator at this Eb/N0 -- that is its property, and the same on both sides.)"""
    from jaero_amd import signalgen as G

    nch, n = 3, 48000 * 5
    rng = np.random.default_rng(21)
    rb = lambda k: bytes(rng.integers(0, 256, k, dtype=np.uint8))
    uw = np.repeat(np.array([(AF.UW >> (31 - k)) & 1 for k in range(32)], dtype=np.uint8), 2)
    sent, pcm = [], np.zeros((nch, n), np.int16)
    for c in range(nch):
        pk = [("R", rb(17)), ("T", (rb(4), [rb(10) for _ in range(3 + c)])), ("R", rb(17))]
        data = [np.concatenate([uw, AF.rt_packet_bits(k, p)]) for k, p in pk]
        pcm[c], _ = G.burst_oqpsk(n, burst_starts=[40000 + 777 * c, 100000 + 555 * c, 185000], ndata_sym=900, fc=8000.0 + 15.0 * c, ebno_db=16.0,
                                  seed=G.SEED_BASE + 300 + c, data=data)
        sent.append(pk)
    chunk = 4096
    demod = B.DemodulatorBank(B.BurstOqpskSettings(), nch, device=0, max_write_samples=chunk, softbit_capacity=16384)
    aerol = B.AeroLBank(nch, 10500, max_softbits_per_write=16384, su_capacity=400, burst=True)
    for s in range(0, n, chunk):
        demod.write(pcm[:, s:s + chunk])
        aerol.write_from_bank(demod, 4096)
    total = 0
    for c in range(nch):
        got = aerol.read_packets(c)
        soft = oracle_mod.run_burst(oracle_mod.burst_oqpsk_settings(), pcm[c], chunk=chunk)["soft"]
        want = oracle_mod.packets_from_rows(oracle_mod.run_aerol_burst(10500, soft)["packets"])
        assert got == want, c
        k0 = 0
        for typ, data in got:  # each decoded packet is the next transmitted one of its kind that it equals
            hits = [k for k in range(k0, len(sent[c])) if (typ == 1 and sent[c][k][0] == "R" and data[:17] == sent[c][k][1])
                    or (typ == 2 and sent[c][k][0] == "T" and data[:4] == sent[c][k][1][0]
                        and [data[6 + 12 * j: 16 + 12 * j] for j in range(len(sent[c][k][1][1]))] == sent[c][k][1][1])]
            assert hits, (c, typ)
            k0 = hits[0] + 1
        total += len(got)
    assert total >= 5
    demod.close()
    aerol.close()
