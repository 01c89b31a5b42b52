"""CPU: the reference's own FFT golden vectors (JAERO/tests/fftwrapper_tests.cpp:27-29, fftrwrapper_tests.cpp:28-30,
"obtained from v1.0.4.11") pin the FFT conventions the oracle (and the JFFT shim of oracle/_ref) use.
Tolerance 1e-5 is the reference tests' own doubles_equal_threshold."""
import numpy as np

from conftest import load_golden

TOL = 0.00001


def close(a, b, tol=TOL):
    """DOUBLES_EQUAL on real and imaginary parts separately, as the reference tests do."""
    a = np.asarray(a, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    return bool(np.all(np.abs(a.real - b.real) <= tol + 1e-12) and np.all(np.abs(a.imag - b.imag) <= tol + 1e-12))


def test_complex_fft_conventions(oracle_mod):
    O = oracle_mod
    g = load_golden("fft_golden")
    x = g["c_input"].astype(np.complex128).copy()
    O.lib().jo_fft(x.ctypes.data, 16, 0)
    assert close(x, g["c_forward"])
    O.lib().jo_fft(x.ctypes.data, 16, 1)  # FFTWrapper inverse: kissfft scaling, i.e. N * input
    assert close(x, g["c_fb"])
    assert close(g["c_fb"], 16 * g["c_input"], 2e-5)


def test_real_fft_conventions(oracle_mod):
    """FFTrWrapper: forward = lower half spectrum (upper half zeroed); the complex oracle FFT reproduces its lower half."""
    O = oracle_mod
    g = load_golden("fft_golden")
    x = g["r_input"].astype(np.complex128).copy()
    O.lib().jo_fft(x.ctypes.data, 16, 0)
    assert close(x[:9], g["r_forward"][:9])
    assert np.all(g["r_forward"][9:] == 0)
    assert close(g["r_fb"], 16 * g["r_input"], 2e-5)


def test_ref_shim_against_golden(oracle_mod):
    """The JFFT stand-in used to build oracle/_ref must pass the same vectors (only where _ref exists)."""
    import pytest

    O = oracle_mod
    if not O.have_ref():
        pytest.skip("oracle/_ref not available")
    g = load_golden("fft_golden")
    try:
        fwd = O.ref_tool("fft", g["c_input"].astype(np.complex128), np.complex128, n=16)
    except Exception as e:
        pytest.skip(f"_ref cannot run here: {e}")
    assert close(fwd, g["c_forward"])
    back = O.ref_tool("ifft", fwd, np.complex128, n=16)
    assert close(back, g["c_fb"])
    rf = O.ref_tool("fftr", g["r_input"].astype(np.float64), np.complex128, n=16)
    assert close(rf, g["r_forward"])
    rb = O.ref_tool("ifftr", rf, np.float64, n=16)
    assert close(rb, g["r_fb"])
