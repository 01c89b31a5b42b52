"""CPU: the C-ABI library loads, exports every symbol include/jaero_hip.h declares, fails loudly without a GPU, and
its host-side scheduling (where the coarse estimate fires) matches the reference's counters."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from jaero_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "jaero_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(jaero_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    L = capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/jaero_hip.h but not exported"
    assert declared == set(capi.EXPORTS)
    assert L.jaero_abi_version() == 1


def test_struct_layout_matches_header():
    assert C.sizeof(capi.Settings) == 48  # 2 ints + 5 doubles
    assert C.sizeof(capi.Status) == 40


def test_no_cpu_fallback():
    """Without a HIP device jaero_create must fail with ENODEV (never silently compute on the host)."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    L = capi.lib()
    s = capi.Settings(capi.KIND_OQPSK, 14, 8000.0, 10500.0, 10500.0, 48000.0, 0.65)
    h = C.c_void_p()
    rc = L.jaero_create(0, 4, C.byref(s), 0, 0, 4096, 0, C.byref(h))
    assert rc == capi.E_NODEV and not h.value
    assert b"fallback" in L.jaero_last_error() or b"device" in L.jaero_last_error()
    assert L.jaero_strerror(rc) == b"no usable HIP device"


def test_ingest_and_aerol_argument_checks_need_no_device():
    """Null / out-of-range arguments are refused before any HIP call (codes and messages as the header documents)."""
    L = capi.lib()
    h = C.c_void_p()
    assert L.jaero_ingest_create(None, 4096, 0, C.byref(h)) == capi.E_INVAL and not h.value
    assert L.jaero_ingest_push(None, 0, b"", 0, 48000) == capi.E_INVAL
    assert L.jaero_ingest_queued(None, 0) == capi.E_INVAL
    assert L.jaero_ingest_pump(None, 0, None, None) == capi.E_INVAL
    assert L.jaero_ingest_stats(None, None) == capi.E_INVAL
    assert b"jaero_ingest" in L.jaero_last_error()
    assert L.jaero_write(None, None, 0, 0, 0, None) == capi.E_INVAL


def _oracle_triggers(O, kind, cpu_reduce, writes):
    d = O.Demod(O.oqpsk_settings() if kind == "oqpsk" else O.msk_settings(), cpu_reduce=cpu_reduce)
    trig, base = [], 0
    rng = np.random.default_rng(0)
    for n in writes:
        x = rng.integers(-2000, 2000, n).astype(np.int16)
        # feed sample by sample so the estimate can be attributed to a sample index
        for i in range(n):
            d.write(x[i:i + 1])
            rows = d.take_status()
            if len(rows):
                trig.append(base + i)
        base += n
    return trig


@pytest.mark.parametrize("kind,power,cpu,writes", [
    ("oqpsk", 14, 0, [4096, 4096, 1000, 777, 5000, 3]),
    ("oqpsk", 14, 1, [30000, 30000, 12345, 40000]),
    ("msk", 13, 0, [2048, 1, 2047, 6000]),
])
def test_schedule_matches_reference_counters(oracle_mod, kind, power, cpu, writes):
    L = capi.lib()
    w = np.array(writes, dtype=np.int32)
    out = np.zeros(256, dtype=np.int64)
    nseg = C.c_int(0)
    n = L.jaero_debug_schedule(power, 48000, cpu, w.ctypes.data, len(writes), out.ctypes.data, 256, C.byref(nseg))
    assert n >= 0
    ref = _oracle_triggers(oracle_mod, kind, bool(cpu), writes)
    assert list(out[:n]) == ref
    assert nseg.value >= n


def test_shard_range_matches_the_python_helper():
    """jaero_shard_range (the C ABI's multi-GPU edge operations) = jaero_amd.dist.shard_range: contiguous, covering, in rank order."""
    from jaero_amd import dist

    L = capi.lib()
    for n, w in ((65536, 8), (32768, 8), (4096, 3), (7, 4), (0, 2), (5, 1)):
        prev = 0
        for r in range(w):
            lo, hi = C.c_int(), C.c_int()
            assert L.jaero_shard_range(n, r, w, C.byref(lo), C.byref(hi)) == 0
            assert (lo.value, hi.value) == dist.shard_range(n, r, w) and lo.value == prev
            prev = hi.value
        assert prev == n
    lo, hi = C.c_int(), C.c_int()
    assert L.jaero_shard_range(8, 2, 2, C.byref(lo), C.byref(hi)) != 0


def _brute_force_triggers(nfft, Fs, flags0, writes, events):
    """Every channel on its own, sample by sample, exactly as the reference's counters move (JAERO/oqpskdemodulator.cpp:410-431):
    ring write when coarseCounter >= Fs or not cpuReduce; fire when the ring pointer reaches a multiple of nfft (cpuReduce) or nfft/4."""
    nch = len(flags0)
    flags = np.array(flags0, dtype=np.int64)
    bb = np.zeros(nch, dtype=np.int64)
    cnt = np.zeros(nch, dtype=np.int64)
    out = []
    base = 0
    for w, ns in enumerate(writes):
        for (bw, ch, kind, val) in events:
            if bw != w:
                continue
            sel = slice(None) if ch < 0 else slice(ch, ch + 1)
            if kind == 0:
                flags[sel] = (flags[sel] & 8) | (val & 7)
            elif kind == 1:
                flags[sel] = (flags[sel] & ~8) | (8 if val else 0)
            else:
                bb[sel] = 0
                cnt[sel] = 0
        for i in range(ns):
            cpu = (flags & 4) != 0
            fill = (cnt >= Fs) | ~cpu
            bb = np.where(fill, (bb + 1) % nfft, bb)
            fire = fill & (bb % np.where(cpu, nfft, nfft // 4) == 0)
            for c in np.nonzero(fire)[0]:
                out.append((base + i, int(c)))
            cnt = np.where(fire, 0, cnt) + 1
        base += ns
    return out


@pytest.mark.parametrize("nch,seed", [(64, 1), (130, 2), (7, 3)])
def test_schedule_with_per_lane_flags_matches_brute_force(nch, seed):
    """The host scheduler with channels of one bank drawing AFC / SQL / cpuReduce independently, flags toggled and setSettings called for
    single channels between writes of odd sizes: which channel fires at which sample, against a per-channel brute-force simulation
    (mixed cpuReduce inside one wavefront had never run anywhere)."""
    L = capi.lib()
    rng = np.random.default_rng(seed)
    power, Fs = 9, 3000  # short cycles: 512-entry ring, the cpuReduce gate opens after 3000 samples
    flags0 = rng.integers(0, 8, nch).astype(np.int32)  # AFC | SQL | cpuReduce in any combination
    writes = [int(x) for x in rng.integers(1, 1500, 40)]
    events = []
    for w in sorted(rng.integers(1, len(writes), 24)):
        kind = int(rng.integers(0, 3))
        ch = int(rng.integers(-1, nch)) if rng.random() < 0.9 else -1
        events.append((int(w), ch, kind, int(rng.integers(0, 8)) if kind == 0 else int(rng.integers(0, 2))))
    ev = np.array(events, dtype=np.int32).reshape(-1, 4)
    w = np.array(writes, dtype=np.int32)
    cap = 1 << 16
    out = np.zeros((cap, 2), dtype=np.int64)
    nseg = C.c_int(0)
    n = L.jaero_debug_schedule_lanes(power, Fs, nch, flags0.ctypes.data, w.ctypes.data, len(writes), ev.ctypes.data, len(events),
                                     out.ctypes.data, cap, C.byref(nseg))
    assert 0 < n <= cap
    got = sorted((int(a), int(b)) for a, b in out[:n])
    want = sorted(_brute_force_triggers(1 << power, Fs, flags0, writes, events))
    assert got == want
    both = {bool(f & 4) for f in flags0}
    assert both == {True, False} and nseg.value >= len({s for s, _ in want})
