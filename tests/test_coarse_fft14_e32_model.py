"""CPU model of the 32 x 32 x 16 workgroup transform of k_coarse6 (jaero_amd/csrc/k_coarse6.h: wg_fft14_e32): two LDS exchanges per
2^14-point transform instead of three.  n = 512 n1 + 16 n2 + n3, k = k1 + 32 k2 + 1024 k3; natural distribution on entry and exit (slot =
index >> 9, thread = index & 511).  A 32-point register FFT leaves its outputs in SPLIT order (slot j < 16 holds X[2j], slot 16 + j holds
X[2j + 1]: one radix-2 stage, then a 16-point FFT on each half in place), which only changes the compile-time addresses of the next
exchange.  Checked: the maps are permutations, the model equals numpy's FFT, every wavefront access is conflict-free: a 64-bit LDS access
is served 16 lanes at a time from 16 eight-byte bank pairs, so every group of 16 consecutive lanes must touch 16 addresses that differ
modulo 16 (the rule SQ_LDS_BANK_CONFLICT confirmed on an MI355X: a first version that only satisfied a 32-lane / 32-bank rule, with row
stride 514 in exchange 2, counted 403 M conflict cycles per launch)."""
import numpy as np

N = 16384
T = np.arange(512)
TW = np.exp(-2j * np.pi * np.arange(N) / N)
XLEN = 16448


def K(s):
    """index held by slot s after a split-order 32-point FFT"""
    return 2 * s if s < 16 else 2 * (s - 16) + 1


def fft32_split(d):
    o = np.fft.fft(d, axis=1)
    return np.stack([o[:, K(s)] for s in range(32)], axis=1)


def ex1_write(s):   # pass-1 thread t = n & 511 = 16 n2 + n3 holds k1 = K(s); odd k1 rows are rotated by one n2 row (16 doubles)
    k1 = K(s)
    return k1 * 512 + ((T + 16 * (k1 & 1)) & 511)


def ex1_read(m):    # pass-2 thread (k1 = t >> 4, n3 = t & 15) wants slot n2 = m: compile-time offset 16 m from a per-thread base (one wrap)
    k1, n3 = T >> 4, T & 15
    return k1 * 512 + ((m + (k1 & 1)) & 31) * 16 + n3


def ex2_write(s):   # pass-2 thread (k1, n3) holds k2 = K(s)
    k1, n3, k2 = T >> 4, T & 15, K(s)
    return (k2 & 15) * 32 + k1 + n3 * 513 + (k2 >> 4) * 8208


def ex2_read(n3, k2hi):   # pass-3 thread t = k & 511 = (k2 & 15) * 32 + k1
    return T + n3 * 513 + k2hi * 8208


def model_fft(x):
    d = x.reshape(32, 512).T.copy()                               # d[t, slot] = x[slot*512 + t]
    o = fft32_split(d)
    o = o * np.stack([TW[T] ** K(s) for s in range(32)], axis=1)  # W_N^(k1 (n mod 512))
    L = np.full(N, np.nan, complex)
    for s in range(32):
        L[ex1_write(s)] = o[:, s]
    d2 = np.stack([L[ex1_read(m)] for m in range(32)], axis=1)
    o = fft32_split(d2)
    o = o * np.stack([TW[32 * (T & 15)] ** K(s) for s in range(32)], axis=1)   # W_512^(k2 n3)
    L = np.full(XLEN, np.nan, complex)
    for s in range(32):
        L[ex2_write(s)] = o[:, s]
    out = np.zeros((512, 32), complex)
    for k2hi in range(2):
        d3 = np.stack([L[ex2_read(n3, k2hi)] for n3 in range(16)], axis=1)
        o3 = np.fft.fft(d3, axis=1)
        for k3 in range(16):
            out[:, 2 * k3 + k2hi] = o3[:, k3]
    return out.T.reshape(N)                                       # out[t, slot] = X[slot*512 + t]


def test_maps_are_permutations():
    assert sorted(np.concatenate([ex1_write(s) for s in range(32)])) == list(range(N))
    assert sorted(np.concatenate([ex1_read(m) for m in range(32)])) == list(range(N))
    w = np.concatenate([ex2_write(s) for s in range(32)])
    r = np.concatenate([ex2_read(n3, h) for n3 in range(16) for h in range(2)])
    assert len(set(w.tolist())) == N and sorted(w) == sorted(r) and w.max() < XLEN


def test_transform_is_natural_in_natural_out():
    rng = np.random.default_rng(32)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    assert np.max(np.abs(model_fft(x) - np.fft.fft(x))) < 1e-8


def test_lds_accesses_are_conflict_free():
    maps = [(f"ex1_write{s}", ex1_write(s)) for s in range(32)] + [(f"ex1_read{m}", ex1_read(m)) for m in range(32)]
    maps += [(f"ex2_write{s}", ex2_write(s)) for s in range(32)] + [(f"ex2_read{n3}{h}", ex2_read(n3, h)) for n3 in range(16) for h in range(2)]
    for name, a in maps:
        for w in range(8):
            addr = a[64 * w:64 * w + 64]
            for q in range(4):
                assert len(set((addr[16 * q:16 * q + 16] % 16).tolist())) == 16, (name, w, q)


# ---------------------------------------------------------------------------------------------------------------------------------------
# wg_fft13_e32 (k_coarse6_13, the MSK rates): 8192 = 32 x 16 x 16 on 256 threads, n = 256 n1 + 16 n2 + n3, k = k1 + 32 k2 + 512 k3; natural in
# and out (slot = index >> 8, thread = index & 255).  Pass 1 as above; pass 2 and pass 3 are two 16-point FFTs each (natural order out).
N13 = 8192
T13 = np.arange(256)
TW13 = np.exp(-2j * np.pi * np.arange(N13) / N13)
XLEN13 = 8208


def e13_ex1_write(s):        # pass-1 thread t = 16 n2 + n3 holds k1 = K(s)
    return K(s) * 256 + T13


def e13_ex1_read(m):         # pass-2 thread (k1a = t >> 4, n3 = t & 15), slot m = 16 g + n2: k1 = k1a + 16 g
    return (m >> 4) * 4096 + (m & 15) * 16 + (T13 >> 4) * 256 + (T13 & 15)


def e13_ex2_write(s):        # pass-2 thread (k1a, n3) holds slot s = 16 g + k2: L = k1 + 32 k2 + 513 n3
    return (s >> 4) * 16 + (s & 15) * 32 + (T13 >> 4) + 513 * (T13 & 15)


def e13_ex2_read(m):         # pass-3 thread t3 = k1 + 32 k2lo, slot m = 16 h + n3: k2 = k2lo + 8 h
    return (m >> 4) * 256 + (m & 15) * 513 + T13


def model_fft13(x):
    d = x.reshape(32, 256).T.copy()                                   # d[t, slot] = x[slot*256 + t]
    o = fft32_split(d) * np.stack([TW13[T13] ** K(s) for s in range(32)], axis=1)   # W_N^(k1 (n mod 256))
    L = np.full(N13, np.nan, complex)
    for s in range(32):
        L[e13_ex1_write(s)] = o[:, s]
    d2 = np.stack([L[e13_ex1_read(m)] for m in range(32)], axis=1)
    k16 = np.arange(16)[None, :]
    o = np.concatenate([np.fft.fft(d2[:, 16 * g:16 * g + 16], axis=1) * TW13[32 * (T13 & 15)][:, None] ** k16 for g in range(2)], axis=1)
    L = np.full(XLEN13, np.nan, complex)
    for s in range(32):
        L[e13_ex2_write(s)] = o[:, s]
    d3 = np.stack([L[e13_ex2_read(m)] for m in range(32)], axis=1)
    out = np.zeros((256, 32), complex)
    for h in range(2):
        o3 = np.fft.fft(d3[:, 16 * h:16 * h + 16], axis=1)
        for k3 in range(16):
            out[:, 2 * k3 + h] = o3[:, k3]
    return out.T.reshape(N13)


def test_fft13_maps_transform_and_banks():
    assert sorted(np.concatenate([e13_ex1_write(s) for s in range(32)])) == list(range(N13))
    assert sorted(np.concatenate([e13_ex1_read(m) for m in range(32)])) == list(range(N13))
    w = np.concatenate([e13_ex2_write(s) for s in range(32)])
    r = np.concatenate([e13_ex2_read(m) for m in range(32)])
    assert len(set(w.tolist())) == N13 and sorted(w) == sorted(r) and w.max() < XLEN13
    rng = np.random.default_rng(13)
    x = rng.standard_normal(N13) + 1j * rng.standard_normal(N13)
    assert np.max(np.abs(model_fft13(x) - np.fft.fft(x))) < 1e-8
    maps = [e13_ex1_write(s) for s in range(32)] + [e13_ex1_read(m) for m in range(32)] + [e13_ex2_write(s) for s in range(32)] + [e13_ex2_read(m) for m in range(32)]
    for a in maps:
        for q in range(16):   # 4 wavefronts x 4 groups of 16 lanes
            assert len(set((a[16 * q:16 * q + 16] % 16).tolist())) == 16


# ---------------------------------------------------------------------------------------------------------------------------------------
# wg_fft14_e64 (scripts/ubench/k_coarse7.h, an experiment that did not reach the library: its 64-point pass does not fit 256 registers):
# 16384 = 64 x 16 x 16 on 256 threads with 64 points each, so that TWO workgroups share a CU (64 KiB of LDS each:
# every exchange moves a plane in two halves).  n = 256 n1 + 16 n2 + n3, k = k1 + 64 k2 + 1024 k3; natural in and out (slot = index >> 8,
# thread = index & 255).  A 64-point register FFT = a radix-4 stage + four 16-point FFTs; its outputs stay in SPLIT-4 order: slot 16 m + q
# holds X[4 q + m].
N64 = 16384
T64 = np.arange(256)
TW64 = np.exp(-2j * np.pi * np.arange(N64) / N64)


def K4(s):
    """index held by slot s after a split-4-order 64-point FFT"""
    return 4 * (s & 15) + (s >> 4)


def fft64_split(d):
    o = np.fft.fft(d, axis=1)
    return np.stack([o[:, K4(s)] for s in range(64)], axis=1)


def e64_ex1_write(k1):       # half = k1 >> 5; pass-1 thread t = 16 n2 + n3
    return (k1 & 31) * 256 + T64


def e64_ex1_read(g, n2):     # pass-2 thread (k1a = t >> 4, n3 = t & 15), group g: k1 = k1a + 16 g (half g >> 1)
    return ((T64 >> 4) + 16 * (g & 1)) * 256 + 16 * n2 + (T64 & 15)


def e64_ex2_write(g, k2):    # pass-2 thread (k1a, n3) holds (g, k2): L = k1 + 64 (k2 & 7) + 513 n3, half = k2 >> 3
    return (T64 >> 4) + 16 * g + 64 * (k2 & 7) + 513 * (T64 & 15)


def e64_ex2_read(h, n3):     # pass-3 thread t3 = k1 + 64 k2lo, group h: k2 = k2lo + 4 h (half h >> 1)
    return T64 + 256 * (h & 1) + 513 * n3


def model_fft64(x):
    d = x.reshape(64, 256).T.copy()                                   # d[t, slot] = x[slot*256 + t]
    o = fft64_split(d) * np.stack([TW64[T64] ** K4(s) for s in range(64)], axis=1)   # W_N^(k1 (n mod 256)), slot s holds k1 = K4(s)
    d2 = np.zeros((256, 64), complex)                                 # slot 16 g + n2
    for half in range(2):
        L = np.full(8192, np.nan, complex)
        for s in range(64):
            if K4(s) >> 5 == half:
                L[e64_ex1_write(K4(s))] = o[:, s]
        for g in range(4):
            if g >> 1 == half:
                for n2 in range(16):
                    d2[:, 16 * g + n2] = L[e64_ex1_read(g, n2)]
    k16 = np.arange(16)[None, :]
    o = np.concatenate([np.fft.fft(d2[:, 16 * g:16 * g + 16], axis=1) * TW64[64 * (T64 & 15)][:, None] ** k16 for g in range(4)], axis=1)  # W_256^(k2 n3)
    d3 = np.zeros((256, 64), complex)                                 # slot 16 h + n3
    for half in range(2):
        L = np.full(8208, np.nan, complex)
        for g in range(4):
            for k2 in range(16):
                if k2 >> 3 == half:
                    L[e64_ex2_write(g, k2)] = o[:, 16 * g + k2]
        for h in range(4):
            if h >> 1 == half:
                for n3 in range(16):
                    d3[:, 16 * h + n3] = L[e64_ex2_read(h, n3)]
    out = np.zeros((256, 64), complex)
    for h in range(4):
        o3 = np.fft.fft(d3[:, 16 * h:16 * h + 16], axis=1)
        for k3 in range(16):
            out[:, 4 * k3 + h] = o3[:, k3]                            # k = t3 + 256 h + 1024 k3 -> natural slot h + 4 k3
    return out.T.reshape(N64)


def test_fft64_maps_transform_and_banks():
    for half in range(2):
        w = np.concatenate([e64_ex1_write(k1) for k1 in range(64) if k1 >> 5 == half])
        r = np.concatenate([e64_ex1_read(g, n2) for g in range(4) if g >> 1 == half for n2 in range(16)])
        assert sorted(w) == list(range(8192)) and sorted(r) == list(range(8192))
        w = np.concatenate([e64_ex2_write(g, k2) for g in range(4) for k2 in range(16) if k2 >> 3 == half])
        r = np.concatenate([e64_ex2_read(h, n3) for h in range(4) if h >> 1 == half for n3 in range(16)])
        assert len(set(w.tolist())) == 8192 and sorted(w) == sorted(r) and w.max() < 8208
    rng = np.random.default_rng(64)
    x = rng.standard_normal(N64) + 1j * rng.standard_normal(N64)
    assert np.max(np.abs(model_fft64(x) - np.fft.fft(x))) < 1e-8
    maps = [e64_ex1_write(k1) for k1 in range(64)] + [e64_ex1_read(g, n2) for g in range(4) for n2 in range(16)]
    maps += [e64_ex2_write(g, k2) for g in range(4) for k2 in range(16)] + [e64_ex2_read(h, n3) for h in range(4) for n3 in range(16)]
    for a in maps:
        for q in range(16):
            assert len(set((a[16 * q:16 * q + 16] % 16).tolist())) == 16
