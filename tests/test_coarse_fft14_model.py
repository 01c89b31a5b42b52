"""CPU model of the two-stream 2^14-point workgroup transform of the coarse-frequency kernel (scripts/ubench/k_coarse5.h: wg_fft14_2s).

The kernel only runs on the GPU (every bank test compares its estimates with the oracle's); what can be pinned without one is its index
arithmetic: 16384 = 16 x 16 x 16 x 4 (decimation in frequency), n = n1*1024 + n2*64 + n3*4 + 2h + q, k = k1 + 16 k2 + 256 k3 + 4096 k4.
The bit h (bit 1 of the index) is passive in passes 1-3, so the 32 points a thread holds are two independent STREAMS of 16 (h = 0, 1)
until the last pass; while one stream's values travel through LDS the other stream's 16-point FFT runs (scripts/ubench/k_coarse5.h; the product kernel is k_coarse6.h, modelled in test_coarse_fft14_e32_model.py).  Distribution D*
(the same on entry and on exit, so three transforms chain register to register): element e sits in stream e1 (bit 1), slot e >> 10,
thread ((e >> 2) & 255) << 1 | (e & 1).  Checked here: (i) the three exchange maps are permutations of their buffer; (ii) with the
twiddles of the kernel the model equals numpy's FFT, D* in and D* out; (iii) every LDS access of a wavefront is conflict-free (64-bit
accesses are served 16 lanes at a time from 16 eight-byte bank pairs) -- most are 64 consecutive doubles."""
import numpy as np

N = 16384
T = np.arange(512)
TW = np.exp(-2j * np.pi * np.arange(N) / N)
W64 = np.exp(-2j * np.pi * np.arange(64) / 64)
S2 = 257  # row stride of exchange 2 (doubles)


def dstar(e):
    """element index -> (stream, thread, slot)"""
    return (e >> 1) & 1, (((e >> 2) & 255) << 1) | (e & 1), e >> 10


# ---- exchange 1 (per stream and plane): pass-1 thread t1 = n2<<5 | n3<<1 | q holds slot k1; pass-2 thread t2 = k1<<5 | n3<<1 | q wants slot n2
def ex1_write(s):
    return s * 512 + T


def ex1_read(s):
    return (T >> 5) * 512 + s * 32 + (T & 31)


# ---- exchange 2: pass-2 thread t2 holds slot k2; pass-3 thread t3 = q<<8 | k1b1<<7 | k2<<3 | (k1>>2)<<1 | (k1&1) wants slot n3
def ex2_write(s):
    k1 = T >> 5
    K = (((k1 >> 1) & 1) << 7) | (s << 3) | ((k1 >> 2) << 1) | (k1 & 1)
    return K + S2 * (T & 31)


def ex2_read(s):
    return (T & 255) + S2 * (2 * s + (T >> 8))


# ---- exchange 3: pass-3 thread t3 holds slot k3; pass-4 thread t4 = (k3&3)<<7 | (t3 & 127) wants slots (q, k3>>2, k1b1)
def ex3_write(s):
    K7, b1, q = T & 127, (T >> 7) & 1, T >> 8
    return ((((s * 2 + q) * 2 + b1) * 2 + (K7 >> 6)) * 64) + (K7 & 63)


def ex3_read(q, k3hi, b1):
    K7, k3 = T & 127, (k3hi << 2) | (T >> 7)
    return ((((k3 * 2 + q) * 2 + b1) * 2 + (K7 >> 6)) * 64) + (K7 & 63)


def model_fft(x):
    """x: natural-order input; returns X in natural order, computed through the D* distribution and the kernel's maps."""
    e = np.arange(N)
    h, t, s = dstar(e)
    d = np.zeros((2, 512, 16), complex)
    d[h, t, s] = x
    k16 = np.arange(16)[None, :]
    p3 = np.zeros((2, 512, 16), complex)
    for st in range(2):
        # pass 1 + W_16384^(k1 (n mod 1024))
        r1 = ((T >> 1) << 2) | (st << 1) | (T & 1)
        o = np.fft.fft(d[st], axis=1) * TW[r1][:, None] ** k16
        L = np.full(8192, np.nan, complex)
        for k in range(16):
            L[ex1_write(k)] = o[:, k]
        d2 = np.stack([L[ex1_read(k)] for k in range(16)], axis=1)
        # pass 2 + W_1024^(k2 (n mod 64))
        r2 = (((T >> 1) & 15) << 2) | (st << 1) | (T & 1)
        o = np.fft.fft(d2, axis=1) * TW[16 * r2][:, None] ** k16
        L = np.full(31 * S2 + 256, np.nan, complex)
        for k in range(16):
            L[ex2_write(k)] = o[:, k]
        d3 = np.stack([L[ex2_read(k)] for k in range(16)], axis=1)
        # pass 3 + W_64^(k3 n4), n4 = 2h + q, q = t3 >> 8
        n4 = 2 * st + (T >> 8)
        p3[st] = np.fft.fft(d3, axis=1) * W64[(n4[:, None] * k16) & 63]
    # exchange 3 (stream-local) and the radix-4 pass over n4 = (h, q)
    got = np.zeros((2, 2, 4, 2, 512), complex)  # [h][q][k3hi][b1][t4]
    for st in range(2):
        L = np.full(8192, np.nan, complex)
        for k in range(16):
            L[ex3_write(k)] = p3[st][:, k]
        for q in range(2):
            for k3hi in range(4):
                for b1 in range(2):
                    got[st, q, k3hi, b1] = L[ex3_read(q, k3hi, b1)]
    out = np.zeros(N, complex)
    for k3hi in range(4):
        for b1 in range(2):
            v = np.stack([got[n4 >> 1, n4 & 1, k3hi, b1] for n4 in range(4)], axis=1)  # [t4][n4]
            o = np.fft.fft(v, axis=1)                                                   # [t4][k4]
            for k4 in range(4):
                # D* on exit: stream b1, slot k4*4 + k3hi, thread t4
                idx = ((k4 * 4 + k3hi) << 10) | ((T >> 1) << 2) | (b1 << 1) | (T & 1)
                out[idx] = o[:, k4]
    return out


def test_exchange_maps_are_permutations():
    assert sorted(np.concatenate([ex1_write(s) for s in range(16)])) == list(range(8192))
    assert sorted(np.concatenate([ex1_read(s) for s in range(16)])) == list(range(8192))
    w2 = np.concatenate([ex2_write(s) for s in range(16)])
    r2 = np.concatenate([ex2_read(s) for s in range(16)])
    assert len(set(w2.tolist())) == 8192 and sorted(w2) == sorted(r2) and w2.max() < 8223
    assert sorted(np.concatenate([ex3_write(s) for s in range(16)])) == list(range(8192))
    r3 = np.concatenate([ex3_read(q, k, b) for q in range(2) for k in range(4) for b in range(2)])
    assert sorted(r3) == list(range(8192))


def test_transform_maps_dstar_to_dstar():
    rng = np.random.default_rng(14)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    X = model_fft(x)
    assert np.max(np.abs(X - np.fft.fft(x))) < 1e-8


def test_dstar_global_accesses_touch_whole_sectors_in_pairs():
    """Ring reads (16-byte elements) and y accesses (8 bytes) of one wavefront in D*: lane pairs are contiguous, the two streams of a slot
    cover the gaps of each other, so a slot's two accesses together touch 64 whole 32-byte (y) / 64-byte (ring) pieces."""
    for w in range(8):
        t = np.arange(64) + 64 * w
        for slot in range(16):
            idx = [(slot << 10) | ((t >> 1) << 2) | (h << 1) | (t & 1) for h in range(2)]
            both = np.sort(np.concatenate(idx))
            assert np.array_equal(both, np.arange(both[0], both[0] + 128))


def test_lds_accesses_are_conflict_free():
    maps = [("ex1_write", ex1_write), ("ex1_read", ex1_read), ("ex2_write", ex2_write), ("ex2_read", ex2_read), ("ex3_write", ex3_write)]
    maps += [(f"ex3_read{q}{k}{b}", (lambda s, q=q, k=k, b=b: ex3_read(q, k, b))) for q in range(2) for k in range(4) for b in range(2)]
    for name, fn in maps:
        for s in range(16 if not name.startswith("ex3_read") else 1):
            a = fn(s)
            for w in range(8):
                addr = a[64 * w:64 * w + 64]
                for q in range(4):  # served 16 lanes at a time from 16 eight-byte bank pairs (SQ_LDS_BANK_CONFLICT agrees: tests/test_coarse_fft14_e32_model.py)
                    assert len(set((addr[16 * q:16 * q + 16] % 16).tolist())) == 16, (name, s, w, q)
                if name not in ("ex2_write", "ex1_read"):
                    assert sorted(addr.tolist()) == list(range(addr.min(), addr.min() + 64)), (name, s, w)
