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


@pytest.mark.parametrize("name", ["a", "b"])
def test_against_reference_golden(B, oracle_mod, name):
    g = load_golden(f"aerol_burst_10500_{name}")
    bank = B.AeroLBank(1, 10500, max_softbits_per_write=4000, su_capacity=600, burst=True)
    feed(bank, [g["soft"]], 4000)
    assert np.array_equal(burst_rows(bank.read_packets(0)), g["packets"])
    ev = bank.read_events(0)
    assert int((ev[:, 1] == 3).sum()) == int(g["bad"])
    starts = np.array([s for s, _ in oracle_mod.demod_groups(g["soft"])])
    dcd = [(int(v), int(starts[np.searchsorted(starts, i, side="right") - 1])) for i, k, v in ev[1:] if k == 0]
    assert dcd == [tuple(r) for r in g["dcd"].tolist()]
    bank.close()


def test_bank_vs_oracle(B, oracle_mod):
    """70 channels (two wave groups): different packets, noise levels, arm inversions, lost tails / late unique words, ragged write
    sizes (so trial lengths, markers and group ends fall anywhere relative to the writes)."""
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
    bank = B.AeroLBank(nch, 10500, max_softbits_per_write=3000, su_capacity=700, burst=True)
    feed(bank, streams, 3000, rng)
    npk = 0
    for c in range(nch):
        o = oracle_mod.run_aerol_burst(10500, streams[c])
        want = oracle_mod.packets_from_rows(o["packets"])
        assert bank.read_packets(c) == want, c
        assert np.array_equal(bank.read_events(c), o["events"]), c
        npk += len(want)
    assert npk > 150
    bank.close()
