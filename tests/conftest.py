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
