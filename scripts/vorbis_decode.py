#!/usr/bin/env python3
"""A small Ogg Vorbis I decoder (floor 1, residues 0 / 1 / 2, channel coupling) -- test infrastructure only.

The reference ships one of its sample recordings as Ogg Vorbis (samples/10.5k_sample.ogg) and this image has no audio decoder of any
kind, so the fixture generator (tests/golden/make_recording_golden.py) decodes it with this file.  Written from the Vorbis I specification;
not bit-exact with libvorbis (the inverse-dB table is computed, not copied) and it does not need to be: the decoded PCM is only the COMMON
INPUT that the unmodified reference (oracle/_ref) and the HIP path are both fed.

    python scripts/vorbis_decode.py in.ogg out.f32 [max_seconds]
"""
import math
import struct
import sys

import numpy as np


# ------------------------------------------------------------------------------------------------ Ogg
def ogg_packets(data):
    """Yields (packet bytes, granule position of the page the packet ends on or -1)."""
    pos, cur = 0, b""
    while pos + 27 <= len(data):
        if data[pos:pos + 4] != b"OggS":
            raise ValueError("lost Ogg sync at %d" % pos)
        _ver, _flags, gp, _serial, _seq, _crc, nseg = struct.unpack_from("<BBqIIIB", data, pos + 4)
        segs = data[pos + 27:pos + 27 + nseg]
        off = pos + 27 + nseg
        ends = []
        for s in segs:
            cur += data[off:off + s]
            off += s
            if s < 255:
                ends.append(cur)
                cur = b""
        for k, pk in enumerate(ends):
            yield pk, (gp if k == len(ends) - 1 else -1)
        pos = off


class EndOfPacket(Exception):
    pass


class BitReader:
    """LSB-first bit reader over one packet (Vorbis I, section 2)."""

    def __init__(self, data):
        self.v = int.from_bytes(data, "little")
        self.n = 8 * len(data)
        self.p = 0

    def read(self, bits):
        if bits == 0:
            return 0
        if self.p + bits > self.n:
            self.p = self.n
            raise EndOfPacket()
        r = (self.v >> self.p) & ((1 << bits) - 1)
        self.p += bits
        return r


def ilog(x):
    return x.bit_length() if x > 0 else 0


def float32_unpack(x):
    mant = x & 0x1FFFFF
    if x & 0x80000000:
        mant = -mant
    exp = (x & 0x7FE00000) >> 21
    return mant * 2.0 ** (exp - 788)


def lookup1_values(entries, dims):
    r = int(math.floor(entries ** (1.0 / dims)))
    while (r + 1) ** dims <= entries:
        r += 1
    while r ** dims > entries:
        r -= 1
    return r


# ------------------------------------------------------------------------------------------------ codebooks
class Codebook:
    def __init__(self, br):
        if br.read(24) != 0x564342:
            raise ValueError("codebook sync")
        self.dims = br.read(16)
        self.entries = br.read(24)
        lengths = [0] * self.entries
        if br.read(1):  # ordered
            cur, ln = 0, br.read(5) + 1
            while cur < self.entries:
                num = br.read(ilog(self.entries - cur))
                for e in range(cur, cur + num):
                    lengths[e] = ln
                cur += num
                ln += 1
        else:
            sparse = br.read(1)
            for e in range(self.entries):
                if sparse:
                    lengths[e] = br.read(5) + 1 if br.read(1) else 0
                else:
                    lengths[e] = br.read(5) + 1
        self.lookup_type = br.read(4)
        self.vq = None
        if self.lookup_type in (1, 2):
            minimum = float32_unpack(br.read(32))
            delta = float32_unpack(br.read(32))
            value_bits = br.read(4) + 1
            sequence_p = br.read(1)
            nvals = lookup1_values(self.entries, self.dims) if self.lookup_type == 1 else self.entries * self.dims
            mult = [br.read(value_bits) for _ in range(nvals)]
            vq = np.zeros((self.entries, self.dims))
            for e in range(self.entries):
                if lengths[e] == 0:
                    continue
                last, div = 0.0, 1
                for i in range(self.dims):
                    off = (e // div) % nvals if self.lookup_type == 1 else e * self.dims + i
                    val = mult[off] * delta + minimum + last
                    vq[e, i] = val
                    if sequence_p:
                        last = val
                    if self.lookup_type == 1:
                        div *= nvals
            self.vq = vq
        elif self.lookup_type != 0:
            raise ValueError("codebook lookup type %d" % self.lookup_type)
        # Huffman: every entry takes the lowest-valued free codeword of its length (the leftmost free leaf at that depth); codewords are
        # read MSb of the codeword first.  table[(length, code)] = entry.
        self.table = {}
        used = [e for e in range(self.entries) if lengths[e] > 0]
        self.single = used[0] if len(used) == 1 else None
        avail = [0] * 33  # avail[l]: next free code at depth l as a left-aligned 32-bit prefix, 0 = none
        first = True
        for e in used:
            ln = lengths[e]
            if first:
                code = 0
                for l in range(1, ln + 1):
                    avail[l] = 1 << (32 - l)
                first = False
                self.table[(ln, 0)] = e
                continue
            z = ln
            while z > 0 and not avail[z]:
                z -= 1
            if z == 0:
                raise ValueError("over-subscribed Huffman tree")
            res = avail[z]
            avail[z] = 0
            self.table[(ln, res >> (32 - ln))] = e
            for y in range(ln, z, -1):
                avail[y] = res + (1 << (32 - y))
        self.maxlen = max(lengths) if used else 0

    def decode_scalar(self, br):
        if self.single is not None:
            br.read(1)
            return self.single
        code, ln, tab = 0, 0, self.table
        while True:
            code = (code << 1) | br.read(1)
            ln += 1
            e = tab.get((ln, code))
            if e is not None:
                return e
            if ln > self.maxlen:
                raise ValueError("bad Huffman code")


# ------------------------------------------------------------------------------------------------ setup
class Floor1:
    def __init__(self, br):
        nparts = br.read(5)
        self.part_class = [br.read(4) for _ in range(nparts)]
        ncls = max(self.part_class) + 1 if nparts else 0
        self.cdim, self.csub, self.cmaster, self.cbooks = [], [], [], []
        for _ in range(ncls):
            self.cdim.append(br.read(3) + 1)
            sub = br.read(2)
            self.csub.append(sub)
            self.cmaster.append(br.read(8) if sub else 0)
            self.cbooks.append([br.read(8) - 1 for _ in range(1 << sub)])
        self.mult = br.read(2) + 1
        rangebits = br.read(4)
        self.X = [0, 1 << rangebits]
        for c in self.part_class:
            for _ in range(self.cdim[c]):
                self.X.append(br.read(rangebits))
        self.order = sorted(range(len(self.X)), key=lambda i: self.X[i])
        self.low, self.high = [0] * len(self.X), [0] * len(self.X)
        for i in range(2, len(self.X)):
            lo = hi = None
            for n in range(i):
                if self.X[n] < self.X[i] and (lo is None or self.X[n] > self.X[lo]):
                    lo = n
                if self.X[n] > self.X[i] and (hi is None or self.X[n] < self.X[hi]):
                    hi = n
            self.low[i], self.high[i] = lo, hi


class Residue:
    def __init__(self, br, rtype):
        self.type = rtype
        self.begin, self.end = br.read(24), br.read(24)
        self.psize = br.read(24) + 1
        self.nclass = br.read(6) + 1
        self.classbook = br.read(8)
        cascade = []
        for _ in range(self.nclass):
            low = br.read(3)
            high = br.read(5) if br.read(1) else 0
            cascade.append(high * 8 + low)
        self.books = [[(br.read(8) if (cascade[i] >> j) & 1 else -1) for j in range(8)] for i in range(self.nclass)]


class Mapping:
    def __init__(self, br, channels):
        if br.read(16) != 0:
            raise ValueError("mapping type")
        self.submaps = br.read(4) + 1 if br.read(1) else 1
        self.coupling = []
        if br.read(1):
            for _ in range(br.read(8) + 1):
                self.coupling.append((br.read(ilog(channels - 1)), br.read(ilog(channels - 1))))
        if br.read(2) != 0:
            raise ValueError("mapping reserved")
        self.mux = [br.read(4) for _ in range(channels)] if self.submaps > 1 else [0] * channels
        self.floor, self.residue = [], []
        for _ in range(self.submaps):
            br.read(8)
            self.floor.append(br.read(8))
            self.residue.append(br.read(8))


INV_DB = np.array([1.0649863e-07 * math.exp(i * math.log(1.0 / 1.0649863e-07) / 255.0) for i in range(256)])  # floor1_inverse_dB_table, computed


def render_point(x0, y0, x1, y1, x):
    dy, adx = y1 - y0, x1 - x0
    off = (abs(dy) * (x - x0)) // adx
    return y0 - off if dy < 0 else y0 + off


def render_line(x0, y0, x1, y1, v):
    dy, adx = y1 - y0, x1 - x0
    ady = abs(dy)
    base = int(dy / adx)  # C division: towards zero
    sy = base - 1 if dy < 0 else base + 1
    ady -= abs(base) * adx
    x, y, err, n = x0, y0, 0, len(v)
    if x < n:
        v[x] = y
    for x in range(x0 + 1, min(x1, n)):
        err += ady
        if err >= adx:
            err -= adx
            y += sy
        else:
            y += base
        v[x] = y


class Vorbis:
    def __init__(self, ident, setup):
        if ident[:7] != b"\x01vorbis" or setup[:7] != b"\x05vorbis":
            raise ValueError("not Vorbis headers")
        _ver, self.channels, self.rate, _a, _b, _c, bs = struct.unpack_from("<IBIiiiB", ident, 7)
        self.bs = (1 << (bs & 15), 1 << (bs >> 4))
        br = BitReader(setup[7:])
        self.books = [Codebook(br) for _ in range(br.read(8) + 1)]
        for _ in range(br.read(6) + 1):
            if br.read(16) != 0:
                raise ValueError("time domain transform")
        self.floors = []
        for _ in range(br.read(6) + 1):
            t = br.read(16)
            if t != 1:
                raise NotImplementedError("floor type %d" % t)
            self.floors.append(Floor1(br))
        self.residues = []
        for _ in range(br.read(6) + 1):
            t = br.read(16)
            if t > 2:
                raise ValueError("residue type")
            self.residues.append(Residue(br, t))
        self.mappings = [Mapping(br, self.channels) for _ in range(br.read(6) + 1)]
        self.modes = []
        for _ in range(br.read(6) + 1):
            flag = br.read(1)
            if br.read(16) or br.read(16):
                raise ValueError("mode window / transform type")
            self.modes.append((flag, br.read(8)))
        if br.read(1) != 1:
            raise ValueError("setup framing bit")
        self.prev = None  # (windowed second half of the previous block, its block size)
        self.twiddle = {}

    # ---- floor 1 ----
    def floor1_decode(self, br, fl, n2):
        if not br.read(1):
            return None
        rng = (256, 128, 86, 64)[fl.mult - 1]
        Y = [br.read(ilog(rng - 1)), br.read(ilog(rng - 1))]
        for c in fl.part_class:
            cdim, cbits = fl.cdim[c], fl.csub[c]
            csub = (1 << cbits) - 1
            cval = self.books[fl.cmaster[c]].decode_scalar(br) if cbits else 0
            for _ in range(cdim):
                book = fl.cbooks[c][cval & csub]
                cval >>= cbits
                Y.append(self.books[book].decode_scalar(br) if book >= 0 else 0)
        X = fl.X
        final = [0] * len(X)
        step2 = [False] * len(X)
        final[0], final[1] = Y[0], Y[1]
        step2[0] = step2[1] = True
        for i in range(2, len(X)):
            lo, hi = fl.low[i], fl.high[i]
            pred = render_point(X[lo], final[lo], X[hi], final[hi], X[i])
            val = Y[i]
            highroom, lowroom = rng - pred, pred
            room = 2 * min(highroom, lowroom)
            if val:
                step2[lo] = step2[hi] = step2[i] = True
                if val >= room:
                    final[i] = val - lowroom + pred if highroom > lowroom else pred - val + highroom - 1
                else:
                    final[i] = pred - (val + 1) // 2 if val & 1 else pred + val // 2
            else:
                final[i] = pred
        curve = [0] * n2
        hx, lx = 0, 0
        ly = final[fl.order[0]] * fl.mult
        hy = ly
        for k in fl.order[1:]:
            if step2[k]:
                hy = final[k] * fl.mult
                hx = X[k]
                render_line(lx, ly, hx, hy, curve)
                lx, ly = hx, hy
        if hx < n2:
            render_line(hx, hy, n2, hy, curve)
        return INV_DB[np.clip(np.asarray(curve), 0, 255)]

    # ---- residue ----
    def residue_decode(self, br, res, n2, nch, skip):
        if res.type == 2:
            vec = self._residue_core(br, res, n2 * nch, 1, [all(skip)])[0]
            out = [vec[c::nch].copy() for c in range(nch)]
            return out
        return self._residue_core(br, res, n2, nch, skip)

    def _residue_core(self, br, res, size, nch, skip):
        out = [np.zeros(size) for _ in range(nch)]
        begin, end = min(res.begin, size), min(res.end, size)
        nparts = (end - begin) // res.psize
        cb = self.books[res.classbook]
        cwords = cb.dims
        if nparts <= 0 or all(skip):
            return out
        classes = [[0] * (nparts + cwords) for _ in range(nch)]
        try:
            for pas in range(8):
                pc = 0
                while pc < nparts:
                    if pas == 0:
                        for j in range(nch):
                            if not skip[j]:
                                temp = cb.decode_scalar(br)
                                for i in range(cwords - 1, -1, -1):
                                    classes[j][i + pc] = temp % res.nclass
                                    temp //= res.nclass
                    for _ in range(cwords):
                        if pc >= nparts:
                            break
                        for j in range(nch):
                            if skip[j]:
                                continue
                            book = res.books[classes[j][pc]][pas]
                            if book < 0:
                                continue
                            bk = self.books[book]
                            off = begin + pc * res.psize
                            v = out[j]
                            if res.type == 0:
                                step = res.psize // bk.dims
                                for i in range(step):
                                    e = bk.vq[bk.decode_scalar(br)]
                                    v[off + i:off + i + step * bk.dims:step] += e
                            else:
                                i = 0
                                while i < res.psize:
                                    e = bk.vq[bk.decode_scalar(br)]
                                    v[off + i:off + i + bk.dims] += e
                                    i += bk.dims
                        pc += 1
        except EndOfPacket:
            pass
        return out

    # ---- transform ----
    def imdct(self, X):
        """y[i] = sum_k X[k] cos(2 pi / n (i + 1/2 + n/4)(k + 1/2)), i < n = 2 len(X)  (Vorbis I, section 1.3.2), by one complex FFT."""
        n2 = len(X)
        n = 2 * n2
        key = n
        if key not in self.twiddle:
            k = np.arange(n2)
            self.twiddle[key] = (np.exp(-1j * np.pi * (k + 0.5) * (1.0 + n / 2.0) / n), np.arange(n))
        pre, i = self.twiddle[key]
        # sum_k X[k] cos(2 pi (i + n0)(k + 1/2) / n), n0 = 1/2 + n/4:  Re{ exp(j pi (i + n0) / n) sum_k X[k] exp(j 2 pi n0 k / n) exp(j 2 pi i k / n) }
        n0 = 0.5 + n / 4.0
        a = np.zeros(n, dtype=np.complex128)
        a[:n2] = X * np.exp(2j * np.pi * n0 * np.arange(n2) / n)
        s = np.fft.ifft(a) * n
        return np.real(s * np.exp(1j * np.pi * (i + n0) / n))

    def window(self, n, flag, prev_flag, next_flag):
        b0 = self.bs[0]
        w = np.zeros(n)
        if flag and not prev_flag:
            ls, le, ln = n // 4 - b0 // 4, n // 4 + b0 // 4, b0 // 2
        else:
            ls, le, ln = 0, n // 2, n // 2
        if flag and not next_flag:
            rs, re_, rn = n * 3 // 4 - b0 // 4, n * 3 // 4 + b0 // 4, b0 // 2
        else:
            rs, re_, rn = n // 2, n, n // 2
        i = np.arange(ls, le)
        w[ls:le] = np.sin(np.pi / 2 * np.sin((i - ls + 0.5) / ln * np.pi / 2) ** 2)
        w[le:rs] = 1.0
        i = np.arange(rs, re_)
        w[rs:re_] = np.sin(np.pi / 2 * np.sin((i - rs + 0.5) / rn * np.pi / 2 + np.pi / 2) ** 2)
        return w

    def decode_packet(self, pk):
        """One audio packet -> float array [channels, samples] (empty for the first packet)."""
        br = BitReader(pk)
        if len(pk) == 0 or br.read(1) != 0:
            return None  # an empty packet / not an audio packet
        flag, mapno = self.modes[br.read(ilog(len(self.modes) - 1))]
        n = self.bs[flag]
        prev_flag = next_flag = 0
        if flag:
            prev_flag, next_flag = br.read(1), br.read(1)
        n2 = n // 2
        mp = self.mappings[mapno]
        floors, unused = [], []
        for c in range(self.channels):
            fl = self.floors[mp.floor[mp.mux[c]]]
            try:
                f = self.floor1_decode(br, fl, n2)
            except EndOfPacket:
                f = None
            floors.append(f)
            unused.append(f is None)
        skip = list(unused)
        for mag, ang in mp.coupling:
            if not (skip[mag] and skip[ang]):
                skip[mag] = skip[ang] = False
        spec = [None] * self.channels
        for sm in range(mp.submaps):
            chs = [c for c in range(self.channels) if mp.mux[c] == sm]
            vecs = self.residue_decode(br, self.residues[mp.residue[sm]], n2, len(chs), [skip[c] for c in chs])
            for c, v in zip(chs, vecs):
                spec[c] = v
        for mag, ang in reversed(mp.coupling):
            # inverse coupling (Vorbis I 4.3.5): the angle value moves one of the two channels off the magnitude
            m, a = spec[mag], spec[ang]
            pos_m, pos_a = m > 0, a > 0
            new_a = np.where(pos_m, np.where(pos_a, m - a, m), np.where(pos_a, m + a, m))
            new_m = np.where(pos_m, np.where(pos_a, m, m + a), np.where(pos_a, m, m - a))
            spec[mag], spec[ang] = new_m, new_a
        w = self.window(n, flag, prev_flag, next_flag)
        cur = []
        for c in range(self.channels):
            if unused[c]:
                cur.append(np.zeros(n))
            else:
                cur.append(self.imdct(spec[c] * floors[c]) * w)
        cur = np.asarray(cur)
        out = np.zeros((self.channels, 0))
        if self.prev is not None:
            # from the centre of the previous window to the centre of this one: D = pn / 4 + n / 4 samples.  The previous block's (windowed) second
            # half starts at t = 0, this block's first half ends at t = D; what lies outside the other's span is under a zero or a flat window.
            pv, pn = self.prev
            D = pn // 4 + n // 4
            out = np.zeros((self.channels, D))
            m = min(D, pn // 2)
            out[:, :m] += pv[:, :m]
            off = D - n2
            j0 = max(0, -off)
            out[:, j0 + off:n2 + off] += cur[:, j0:n2]
        self.prev = (cur[:, n2:].copy(), n)
        return out


def decode(path, max_seconds=None):
    data = open(path, "rb").read()
    it = ogg_packets(data)
    ident, _ = next(it)
    _comment, _ = next(it)
    setup, _ = next(it)
    v = Vorbis(ident, setup)
    chunks, total = [], 0
    limit = None if max_seconds is None else int(max_seconds * v.rate)
    for pk, _gp in it:
        o = v.decode_packet(pk)
        if o is None or o.shape[1] == 0:
            continue
        chunks.append(o)
        total += o.shape[1]
        if limit is not None and total >= limit:
            break
    pcm = np.concatenate(chunks, axis=1) if chunks else np.zeros((v.channels, 0))
    if limit is not None:
        pcm = pcm[:, :limit]
    return pcm, v.rate


if __name__ == "__main__":
    pcm, rate = decode(sys.argv[1], float(sys.argv[3]) if len(sys.argv) > 3 else None)
    pcm.T.astype("<f4").tofile(sys.argv[2])
    print("decoded %d samples x %d channels at %d Hz, peak %.3f, rms %.4f" % (pcm.shape[1], pcm.shape[0], rate, np.abs(pcm).max(), np.sqrt((pcm ** 2).mean())))
