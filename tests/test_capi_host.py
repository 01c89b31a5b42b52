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
