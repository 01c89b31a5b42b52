"""Bank-scale flags matrix: the per-object switches of the reference demodulators -- setAFC / setSQL / setCPUReduce
(JAERO/oqpskdemodulator.cpp:149-163, JAERO/mskdemodulator.cpp:105-118), DCDstatSlot (:679-684) and setSettings (:175-289) -- exercised PER
CHANNEL inside one bank.  Every channel draws its three flags independently, and at six moments that are multiples neither of the write size
nor of nfft/4 a channel may get new flags, a DCD change or a setSettings of its own, so that inside one wavefront of 64 channels

  * cpuReduce and non-cpuReduce channels sit side by side (ring fills gated by coarseCounter for some lanes only, estimates every nfft
    samples for some and every nfft/4 for others: Mirror::steps_to_trigger / next_segment / fired in jaero_hip.hip),
  * estimate phases diverge (setSettings restarts a channel's ring pointer at an unaligned sample),
  * AFC retunes mixer_center for some lanes (bigchange + ring cleared) and not for others.

Spread channels are compared with oracle objects that received the same calls between the same writes: soft bits, soft symbols (the small
banks), every status row."""
import numpy as np
import pytest

from test_gpu_parity import compare

pytestmark = pytest.mark.gpu

EVENT_TIMES = [31337, 47003, 70001, 90011, 120007, 141233]  # none a multiple of 4096 (the write size, and nfft/4 at 10.5 kbps) or of 2048
NSAMP = 172000  # 3.58 s: the cpuReduce channels' first estimate needs 48 000 + 16 384 samples
NSIG = 16


def plan(c, kind):
    """Channel c's initial flags and its events [(sample, what, payload)], a pure function of c (the oracle side replays it)."""
    r = np.random.default_rng(7000 + c)
    flags = (bool(r.random() < 0.5), bool(r.random() < 0.5), bool(r.random() < 0.35))
    ev = []
    cur = flags
    for t in EVENT_TIMES:
        u = r.random()
        if u < 0.25:
            cur = (bool(r.random() < 0.5), bool(r.random() < 0.5), bool(r.random() < 0.4))
            ev.append((t, "flags", cur))
        elif u < 0.45:
            ev.append((t, "dcd", bool(r.random() < 0.6)))
        elif u < 0.60:
            if kind == "oqpsk":
                ev.append((t, "settings", (8000.0 + float(r.integers(-40, 41)), float(r.choice([9000.0, 10500.0])))))
            else:
                ev.append((t, "settings", (1000.0 + float(r.integers(-20, 21)), float(r.choice([1500.0, 1800.0])))))
    return flags, ev


def settings_pair(B, O, kind, fc=None, lbw=None):
    if kind == "oqpsk":
        fc, lbw = (8000.0 if fc is None else fc), (10500.0 if lbw is None else lbw)
        return B.OqpskSettings(freq_center=fc, lockingbw=lbw), O.oqpsk_settings(freq_center=fc, lockingbw=lbw)
    fc, lbw = (1000.0 if fc is None else fc), (1800.0 if lbw is None else lbw)
    return B.MskSettings(fb=1200.0, freq_center=fc, lockingbw=lbw), O.msk_settings(fb=1200.0, freq_center=fc, lockingbw=lbw)


_SIG = {}


def signals(kind):
    from jaero_amd import signalgen as G

    if kind in _SIG:
        return _SIG[kind]
    rows = []
    for k in range(NSIG):
        if kind == "oqpsk":
            rows.append(G.oqpsk(NSAMP, fc=8000.0 + 11.0 * (k - NSIG // 2), ebno_db=11.0 + (k % 3), seed=G.SEED_BASE + 3300 + k)[0])
        else:
            rows.append(G.msk(NSAMP, fc=1000.0 + 5.0 * (k - NSIG // 2), ebno_db=11.0 + (k % 3), seed=G.SEED_BASE + 3400 + k, fb=1200.0)[0])
    _SIG[kind] = np.stack(rows)
    return _SIG[kind]


def run_matrix(B, O, kind, nch, check, capture):
    sig = signals(kind)
    src = np.arange(nch) % NSIG
    plans = [plan(c, kind) for c in range(nch)]
    bs, _ = settings_pair(B, O, kind)
    chunk = 4096
    cap = int(NSAMP * (10500 if kind == "oqpsk" else 1200) / 48000) + 64
    bank = B.DemodulatorBank(bs, nch, ebno=True, status_log=True, capture_symbols=capture, max_write_samples=chunk, softbit_capacity=cap)
    for c in range(nch):
        a, s, r = plans[c][0]
        bank.set_flags(a, s, r, channel=c)
    by_time = {t: [] for t in EVENT_TIMES}
    for c in range(nch):
        for (t, what, payload) in plans[c][1]:
            by_time[t].append((c, what, payload))
    cuts = [0] + EVENT_TIMES + [NSAMP]
    nwrites = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        for (c, what, payload) in by_time.get(a, []):
            if what == "flags":
                bank.set_flags(*payload, channel=c)
            elif what == "dcd":
                bank.set_dcd(payload, channel=c)
            else:
                bank.set_settings(settings_pair(B, O, kind, *payload)[0], channel=c)
        for s in range(a, b, chunk):
            e = min(s + chunk, b)
            bank.write(sig[src, s:e])
            nwrites += 1
    seen = {"flags": 0, "dcd": 0, "settings": 0, "cpu": 0, "afc": 0, "soft": 0, "rows": 0}
    for c in check:
        (a0, s0, r0), evs = plans[c]
        d = O.Demod(settings_pair(B, O, kind)[1], afc=a0, sql=s0, cpu_reduce=r0, capture_symbols=capture)
        seen["cpu"] += int(r0); seen["afc"] += int(a0)
        ev = {t: [] for t in EVENT_TIMES}
        for (t, what, payload) in evs:
            ev[t].append((what, payload))
            seen[what] += 1
        for a, b in zip(cuts[:-1], cuts[1:]):
            for (what, payload) in ev.get(a, []):
                if what == "flags":
                    d.set_flags(*payload)
                elif what == "dcd":
                    d.set_dcd(payload)
                else:
                    d.set_settings(settings_pair(B, O, kind, *payload)[1])
            for s in range(a, b, chunk):
                d.write(sig[src[c], s:min(s + chunk, b)])
        ref = {"soft": d.take_soft(), "status": d.take_status(), "pending": d.pending}
        if capture:
            ref["symbols"] = d.take_symbols()
        try:
            compare(bank.read_softbits(c), bank.read_symbols(c) if capture else None, bank.read_status_log(c, caprows=1 << 12), ref)
        except AssertionError as e:
            raise AssertionError(f"channel {c} (flags {plans[c][0]}, events {evs}): {e}") from e
        seen["soft"] += len(ref["soft"]); seen["rows"] += len(ref["status"])
    bank.close()
    return seen, plans


def spread(nch, k=12):
    base = sorted({0, 1, 63, 64, 65, nch // 3, nch // 2, nch - 66, nch - 65, nch - 3, nch - 2, nch - 1} & set(range(nch)))
    return base[:k] if len(base) >= k else sorted(set(base) | set(range(min(nch, k))))


@pytest.fixture(scope="module")
def B():
    from jaero_amd import capi
    from jaero_amd import demodulator as D

    capi.lib()
    return D


def _assert_covered(seen, plans, check):
    # the checked channels between them exercise every switch, on and off, and did lock
    assert seen["flags"] >= 2 and seen["dcd"] >= 2 and seen["settings"] >= 2, seen
    assert 0 < seen["cpu"] < len(check) and 0 < seen["afc"] < len(check), seen
    assert seen["soft"] > 1000 * len(check) // 2 and seen["rows"] > 4 * len(check), seen
    grp = plans[:64]
    assert {p[0][2] for p in grp} == {True, False}  # cpuReduce and non-cpuReduce lanes inside the first wavefront


def test_oqpsk_bank_130_flags_matrix(B, oracle_mod):
    check = spread(130)
    seen, plans = run_matrix(B, oracle_mod, "oqpsk", 130, check, capture=True)
    _assert_covered(seen, plans, check)


def test_msk_bank_130_flags_matrix(B, oracle_mod):
    check = spread(130)
    seen, plans = run_matrix(B, oracle_mod, "msk", 130, check, capture=True)
    _assert_covered(seen, plans, check)


def test_oqpsk_bank_33091_flags_matrix(B, oracle_mod):
    """The same in the four-pair kernel with a ragged last workgroup (518 groups: the last workgroup holds two live pairs, the last group
    three live lanes)."""
    nch = 33091
    check = spread(nch)
    seen, plans = run_matrix(B, oracle_mod, "oqpsk", nch, check, capture=False)
    _assert_covered(seen, plans, check)
