import ast
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    if "opts" in d:
        d["opts"] = ast.literal_eval(str(d["opts"]))
    if "kind" in d:
        d["kind"] = str(d["kind"])
    return d


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O

    O.build()
    return O


def oracle_settings(O, kind, opts):
    if kind == "oqpsk":
        return O.oqpsk_settings(freq_center=opts.get("freq_center", 8000.0), lockingbw=opts.get("lockingbw", 10500.0),
                                fb=opts.get("fb", 10500.0), power=opts.get("power", 14), threshold=opts.get("threshold", 0.65))
    return O.msk_settings(freq_center=opts.get("freq_center", 1000.0), lockingbw=opts.get("lockingbw", 1800.0),
                          fb=opts.get("fb", 1200.0), power=opts.get("power", 13), threshold=opts.get("threshold", 0.5),
                          Fs=float(opts.get("Fs", 48000.0)))


def bank_settings(kind, opts):
    from jaero_amd.demodulator import MskSettings, OqpskSettings

    if kind == "oqpsk":
        return OqpskSettings(freq_center=opts.get("freq_center", 8000.0), lockingbw=opts.get("lockingbw", 10500.0),
                             fb=float(opts.get("fb", 10500.0)), coarsefreqest_fft_power=opts.get("power", 14),
                             signalthreshold=opts.get("threshold", 0.65))
    return MskSettings(freq_center=opts.get("freq_center", 1000.0), lockingbw=opts.get("lockingbw", 1800.0),
                       fb=opts.get("fb", 1200.0), coarsefreqest_fft_power=opts.get("power", 13),
                       signalthreshold=opts.get("threshold", 0.5), Fs=float(opts.get("Fs", 48000.0)))


@pytest.fixture
def force_viterbi_layout():
    """The Viterbi decoder picks its layout by size; tests force one ("wave" / "lanes" / "auto") through the library's test hook
    jaero_debug_viterbi_layout so that both layouts meet the oracle at small sizes.  Reset to "by size" afterwards."""
    from jaero_amd import capi

    def force(name: str):
        capi.check(capi.lib().jaero_debug_viterbi_layout({"auto": 0, "wave": 1, "lanes": 2}[name]))

    yield force
    capi.lib().jaero_debug_viterbi_layout(0)


# ---- soft bytes: counted, not just bounded -------------------------------------------------------------------------------------------
# A soft bit is qRound of a float-derived value clipped to a byte (JAERO/oqpskdemodulator.cpp:583-595): where the fp64 value sits within
# an ulp of a rounding edge the byte may differ by one.  Every GPU test bounds the difference by 1 AND counts the bytes that differ:
# the ledger of a session (test -> compared, differing) is written to gpurun_out/soft_byte_ledger.json, and the count allowed per call is
# what the suite saw when the call site was written (0 unless the call says otherwise) -- a drift from 0 to 0.1 % off-by-one fails.
_SOFT_LEDGER = {}


def assert_soft_bytes(got, ref, where="", allow=0):
    got = np.asarray(got).astype(int)
    ref = np.asarray(ref).astype(int)
    assert got.shape == ref.shape, (where, got.shape, ref.shape)
    d = np.abs(got - ref)
    ndiff = int((d != 0).sum())
    key = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    e = _SOFT_LEDGER.setdefault(key, {"compared": 0, "differing": 0, "max": 0})
    e["compared"] += int(d.size); e["differing"] += ndiff; e["max"] = max(e["max"], int(d.max(initial=0)))
    assert d.max(initial=0) <= 1, (where, "a soft byte differs by more than one", int(d.max()))
    assert ndiff <= allow, (where, f"{ndiff} of {d.size} soft bytes differ (allowed {allow})")
    return ndiff


def pytest_sessionfinish(session, exitstatus):
    if not _SOFT_LEDGER:
        return
    import json

    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        tot = {"compared": sum(e["compared"] for e in _SOFT_LEDGER.values()), "differing": sum(e["differing"] for e in _SOFT_LEDGER.values())}
        with open(os.path.join(out, "soft_byte_ledger.json"), "w") as f:
            json.dump({"total": tot, "tests": _SOFT_LEDGER}, f, indent=1, sort_keys=True)
    except OSError:
        pass
