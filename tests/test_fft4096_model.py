"""CPU model of the 4096-point workgroup transform behind k_pre8400_fft / k_hilbert_fft (jaero_amd/csrc/k_pre8400.h: pf_fft4096).

The kernel itself only runs on the GPU (tests/test_jfastfir_vectors.py, tests/test_gpu_parity.py, tests/test_gpu_burst.py); what can be
pinned without one is its index arithmetic: (i) the two exchange maps and the three twiddle sets turn "thread T holds x[T + 256 s]" into
"thread T holds X[T + 256 s]"; (ii) the overlap-save identity the kernels use (window, valid half, 1/N and the conjugate trick folded
into H); (iii) the LDS addresses of every exchange access: 64 consecutive doubles per wavefront, except the swizzled transposing write,
where each half wavefront must cover all 32 eight-byte banks exactly once."""
import numpy as np

N = 4096
T = np.arange(256)
HI, LO = T >> 4, T & 15
TW = np.exp(-2j * np.pi * np.arange(N) / N)


def ex1_write(s):   # writer T = 16 m1 + m2, slot k1 = s
    return s * 256 + T


def ex1_read(s):    # reader T = 16 k1 + m2, slot m1 = s
    return HI * 256 + s * 16 + LO


def ex2_write(s):   # writer T = 16 k1 + m2, slot q1 = s
    return (s * 16 + LO) * 16 + (HI ^ LO)


def ex2_read(s):    # reader T = k1 + 16 q1, slot m2 = s
    return (HI * 16 + s) * 16 + (LO ^ s)


def model_fft(d):
    """d[T, s] = x[T + 256 s]  ->  X[T + 256 s], following pf_fft4096 step by step."""
    o = np.fft.fft(d, axis=1) * TW[T][:, None] ** np.arange(16)[None, :]           # pass 1 + W_4096^(T k1)
    L = np.zeros(N, complex)
    for s in range(16):
        L[ex1_write(s)] = o[:, s]
    d2 = np.stack([L[ex1_read(s)] for s in range(16)], axis=1)
    o = np.fft.fft(d2, axis=1) * TW[16 * LO][:, None] ** np.arange(16)[None, :]   # pass 2 + W_256^(m2 q1)
    L = np.zeros(N, complex)
    for s in range(16):
        L[ex2_write(s)] = o[:, s]
    d3 = np.stack([L[ex2_read(s)] for s in range(16)], axis=1)
    return np.fft.fft(d3, axis=1)                                                  # pass 3


def test_exchange_maps_are_permutations():
    for w, r in ((ex1_write, ex1_read), (ex2_write, ex2_read)):
        allw = np.concatenate([w(s) for s in range(16)])
        allr = np.concatenate([r(s) for s in range(16)])
        assert sorted(allw) == list(range(N)) and sorted(allr) == list(range(N))


def test_transform_is_natural_in_natural_out():
    rng = np.random.default_rng(3)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    X = model_fft(x.reshape(16, 256).T.copy())
    assert np.max(np.abs(X - np.fft.fft(x).reshape(16, 256).T)) < 1e-9


def test_overlap_save_identity():
    """out[m0 + j] = sum_k h[k] x[m0 + j - 2048 - k] from the window x[m0 - 4096 .. m0 - 1]: window indices 2048 .. 4095 of the circular
    convolution, computed as conj(FFT(conj(FFT(window) * H))) with H = FFT(h, 4096) / 4096 -- slots 8 .. 15 of every thread."""
    rng = np.random.default_rng(4)
    xw = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    h = rng.standard_normal(2049)
    H = np.fft.fft(np.r_[h, np.zeros(N - 2049)]) / N
    X = model_fft(xw.reshape(16, 256).T.copy())
    Y = np.conj(X * H.reshape(16, 256).T)
    y = np.conj(model_fft(Y))                       # y[T, s] = result at window index T + 256 s
    lin = np.convolve(xw, h)
    for s in range(8, 16):
        assert np.max(np.abs(y[:, s] - lin[T + 256 * s])) < 1e-9


def test_two_real_channels_share_a_transform_when_the_taps_are_real():
    """k_hilbert_fft: z = x_a + j x_b, g real  =>  g (*) z = (g (*) x_a) + j (g (*) x_b)."""
    rng = np.random.default_rng(5)
    xa, xb, g = rng.standard_normal(N), rng.standard_normal(N), rng.standard_normal(2048)
    G = np.fft.fft(np.r_[g, np.zeros(N - 2048)]) / N
    Z = model_fft((xa + 1j * xb).reshape(16, 256).T.copy())
    w = np.conj(model_fft(np.conj(Z * G.reshape(16, 256).T)))
    la, lb = np.convolve(xa, g), np.convolve(xb, g)
    for s in range(8, 16):
        assert np.max(np.abs(w[:, s].real - la[T + 256 * s])) < 1e-9
        assert np.max(np.abs(w[:, s].imag - lb[T + 256 * s])) < 1e-9


def test_lds_accesses_are_conflict_free():
    """LDS address (in doubles) = 4 L + c for thread (c = tid & 3, T = tid >> 2); a wavefront = 16 consecutive T x 4 c.  64 banks of 4
    bytes = 32 banks of one double; a 64-bit access is served half a wavefront at a time."""
    for name, fn in (("ex1_write", ex1_write), ("ex1_read", ex1_read), ("ex2_write", ex2_write), ("ex2_read", ex2_read)):
        for s in range(16):
            L = fn(s)
            for w in range(16):                      # wavefront w: T = 16 w .. 16 w + 15
                addr = (4 * L[16 * w:16 * w + 16][:, None] + np.arange(4)[None, :]).reshape(-1)   # lane order: T major, c minor
                for half in (addr[:32], addr[32:]):
                    banks = half % 32
                    assert len(set(banks.tolist())) == 32, (name, s, w)
                if name != "ex2_write":
                    assert sorted(addr.tolist()) == list(range(addr.min(), addr.min() + 64)), (name, s, w)
