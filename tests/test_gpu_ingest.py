"""GPU (-m gpu): batched ingest (jaero_ingest_*) through the C ABI.

The reference hands each ZMQ message to one demodulator: recAudio(QByteArray, sampleRate) -> dataReceived -> writeData
(JAERO/zmq_audioreceiver.cpp:40-79, oqpskdemodulator.cpp:686-693).  Here messages of arbitrary sizes arrive for the
channels of a bank in arbitrary order; what comes out must equal (a) the oracle fed exactly those messages, one
writeData per message, and (b) the same bank fed directly."""
import numpy as np
import pytest

from conftest import assert_soft_bytes, bank_settings, oracle_settings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    from jaero_amd import capi
    from jaero_amd import demodulator as D

    capi.lib()
    return D


def messages(rng, n, lo, hi):
    """Cut n samples into messages of lo..hi samples."""
    cuts, s = [], 0
    while s < n:
        m = int(rng.integers(lo, hi + 1))
        cuts.append((s, min(n, s + m)))
        s += m
    return cuts


@pytest.mark.parametrize("kind,nch,nsamp,chunk", [("oqpsk", 5, 70000, 4096), ("msk", 3, 60000, 3000), ("oqpsk", 66, 30000, 2048)])
def test_messages_vs_oracle(B, oracle_mod, kind, nch, nsamp, chunk):
    from jaero_amd import signalgen as G

    O = oracle_mod
    rng = np.random.default_rng(1234 + nch)
    pcm, _, _ = G.channel_bank(kind, nch, nsamp, ebno_db=11.0, seed0=G.SEED_BASE + 900)
    opts = {} if kind == "oqpsk" else {"fb": 1200.0, "lockingbw": 1800.0}
    bank = B.DemodulatorBank([bank_settings(kind, opts) for _ in range(nch)], ebno=True, status_log=True, capture_symbols=False,
                             max_write_samples=chunk, softbit_capacity=nsamp)
    ing = B.Ingest(bank, chunk, 4 * chunk)
    cuts = [messages(rng, nsamp, 100, 2 * chunk) for _ in range(nch)]
    nxt = [0] * nch
    writes = 0
    while any(nxt[c] < len(cuts[c]) for c in range(nch)):
        order = rng.permutation(nch)
        progressed = False
        for c in order:
            if nxt[c] >= len(cuts[c]):
                continue
            a, b = cuts[c][nxt[c]]
            if ing.queued(int(c)) + (b - a) > 4 * chunk:
                continue  # this channel is ahead of the others: its sender waits
            assert ing.dataReceived(int(c), pcm[c, a:b].astype("<i2").tobytes(), 48000) == 0
            nxt[c] += 1
            progressed = True
        writes += ing.pump()
        assert progressed or ing.queued(-1) > 0
    writes += ing.pump(flush=True)
    st = ing.stats()
    assert st["samples_written"] == nsamp and st["refused_samples"] == 0 and st["rate_warnings"] == 0
    assert writes >= nsamp // chunk
    check = range(nch) if nch <= 8 else sorted({0, 1, 63, 64, nch - 1})
    for c in check:
        # the reference: one writeData per message
        ref = O.run_demod(oracle_settings(O, kind, opts), pcm[c], chunk=[b - a for a, b in cuts[c]], capture_symbols=False)
        soft, log = bank.read_softbits(c), bank.read_status_log(c)
        n = len(ref["soft"])
        assert len(soft) == n + ref["pending"]
        assert np.array_equal(soft[:n] >= 128, ref["soft"] >= 128)
        assert_soft_bytes(soft[:n], ref["soft"])
        assert log.shape == ref["status"].shape
        if len(log):
            assert np.array_equal(log[:, [0, 5]], ref["status"][:, [0, 5]])
            assert np.max(np.abs(log[:, 1:4] - ref["status"][:, 1:4])) < 1e-6
    ing.close()
    bank.close()


def test_equals_direct_writes_and_error_behaviour(B):
    from jaero_amd import capi
    from jaero_amd import signalgen as G

    nch, nsamp, chunk = 4, 40000, 4096
    pcm, _, _ = G.channel_bank("oqpsk", nch, nsamp, ebno_db=10.0, seed0=G.SEED_BASE + 950)
    direct = B.DemodulatorBank(bank_settings("oqpsk", {}), nch, ebno=False, max_write_samples=chunk, softbit_capacity=nsamp)
    for s in range(0, nsamp, chunk):
        direct.write(pcm[:, s:s + chunk])
    bank = B.DemodulatorBank(bank_settings("oqpsk", {}), nch, ebno=False, max_write_samples=chunk, softbit_capacity=nsamp)
    with pytest.raises(capi.JaeroError):
        B.Ingest(bank, chunk + 1)  # a chunk must fit one jaero_write
    ing = B.Ingest(bank, chunk, 3 * chunk)
    # nothing is written until every channel has a chunk
    for c in range(nch - 1):
        ing.dataReceived(c, pcm[c, :chunk].astype("<i2").tobytes())
    assert ing.pump() == 0 and ing.queued(-1) == 0 and ing.queued(0) == chunk
    # an odd trailing byte is ignored (writeData works on len/2 samples); a wrong sample rate is a warning for OQPSK
    assert ing.dataReceived(nch - 1, pcm[nch - 1, :chunk].astype("<i2").tobytes() + b"\x7f", 44100) == capi.W_RATE
    assert ing.pump() == 1
    # overflow: the message is refused whole and can be offered again after a pump
    big = pcm[0, chunk:4 * chunk + 2].astype("<i2").tobytes()
    with pytest.raises(capi.JaeroError) as e:
        ing.dataReceived(0, big)
    assert e.value.code == -5
    assert ing.queued(0) == 0
    with pytest.raises(capi.JaeroError):
        ing.dataReceived(nch, b"")  # bad channel
    for c in range(nch):
        ing.dataReceived(c, pcm[c, chunk:].astype("<i2").tobytes()[: 2 * 3 * chunk])
    ing.pump()
    for s0 in range(4 * chunk, nsamp, 3 * chunk - 7):  # the rest, in messages that do not line up with the slots
        for c in range(nch):
            ing.dataReceived(c, pcm[c, s0:s0 + 3 * chunk - 7].astype("<i2").tobytes())
        ing.pump(flush=(s0 == 4 * chunk))  # one early flush: the read position leaves the slot grid and finds it again
    ing.pump(flush=True)
    st = ing.stats()
    assert st["samples_written"] == nsamp and st["rate_warnings"] == 1 and st["refused_samples"] == 3 * chunk + 2
    for c in range(nch):
        assert np.array_equal(bank.read_softbits(c), direct.read_softbits(c))
    ing.close()
    bank.close()
    direct.close()


def test_msk_rate_change_is_refused(B):
    from jaero_amd import capi

    bank = B.DemodulatorBank(bank_settings("msk", {"fb": 1200.0, "lockingbw": 1800.0}), 2, ebno=False, max_write_samples=1024)
    ing = B.Ingest(bank, 1024)
    with pytest.raises(capi.JaeroError) as e:
        ing.dataReceived(0, bytes(200), 24000)
    assert e.value.code == -6 and ing.queued(0) == 0
    assert ing.dataReceived(0, bytes(200), 48000) == 0 and ing.queued(0) == 100
    ing.close()
    bank.close()
