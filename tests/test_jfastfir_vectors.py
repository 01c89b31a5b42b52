"""The third of the reference's golden-vector tests for the JFFT stand-in: JAERO/tests/jfastfir_tests.cpp:31-58 feeds a
recorded input through JFastFir with the kernel RRC(0.6, 2049 taps, 48 kHz, 5250 sym/s), nfft 4096, and requires the output recorded
from JAERO v1.0.4.11 from sample 4096 on, at 1e-5.  Same vectors (tests/golden/jfastfir.npz, made by make_golden.py jfastfir), same
bar, for: the unmodified JFastFir over the stand-in (oracle/_ref), the oracle's restatement, and the GPU's prefilter kernels -- the
overlap-save FFT form k_pre8400_fft the 8400 bps C-channel path runs (exactly this filter at alpha 0.6) and the direct form
k_pre8400_fir kept beside it."""
import numpy as np
import pytest

from conftest import load_golden

TOL = 1e-5  # doubles_equal_threshold of the reference test


def check(got, g):
    want = g["expected_output"]
    assert len(got) == len(want)
    assert np.max(np.abs(got[4096:] - want[4096:])) < TOL  # "the first 4096 samples need not match"


def test_oracle_restatement(oracle_mod):
    g = load_golden("jfastfir")
    check(oracle_mod.fastfir(g["input"]), g)


def test_unmodified_jfastfir_over_the_shim(oracle_mod):
    if not oracle_mod.have_ref():
        pytest.skip("oracle/_ref/jaero_ref not available here")
    g = load_golden("jfastfir")
    for chunk in (len(g["input"]), 1000):
        got = oracle_mod.ref_tool("fastfir", g["input"], np.complex128, alpha=0.6, K=2048, nfft=4096, Fs=48000, fsym=5250, chunk=chunk)
        check(got, g)


@pytest.mark.gpu
def test_gpu_prefilter_kernel():
    from jaero_amd import capi

    g = load_golden("jfastfir")
    x = np.ascontiguousarray(g["input"], dtype=np.complex128)
    out = np.empty_like(x)
    capi.check(capi.lib().jaero_debug_prefilter(0, x.ctypes.data, len(x), 0.6, 5250.0, out.ctypes.data))
    check(out, g)
    # and against the oracle on the whole length (both sum the same products, in different orders)
    from oracle import oracle as O

    ref = O.fastfir(x)
    assert np.max(np.abs(out[2048:] - ref[2048:])) < 1e-11
