#!/usr/bin/env python3
"""MPEG-1 audio Layer III decoder written from the standard (ISO/IEC 11172-3) -- TEST INFRASTRUCTURE: the image has no audio decoder and the
reference's only off-air recording of a 10.5 kbps R/T (burst) channel is samples/10.5k_burst_sample.mp3 (missing #2).  Used once, in the
build container, by tests/golden/make_burst_recording_golden.py; the PCM it produces is the COMMON input of the unmodified reference, the oracle and the
GPU bank, so the decoder need not be bit-exact with any other decoder (a conforming decoder is within +-1 LSB of the reference decoder; this one
computes in float64).

Covers what the standard's Layer III allows for MPEG-1 streams: mono / stereo / joint stereo with mid-side (intensity stereo is refused: say so
rather than decode it wrong), long / short / mixed blocks, scfsi, the bit reservoir, CRC-protected frames (the CRC is checked).  Self-checks,
reported in `info` and asserted by the caller:
  * every frame header is in sync at the position the previous frame's length gives (no resynchronisation is ever needed);
  * the bit reservoir closes on every frame: main_data_begin never points before the data that exists, and a granule's data never extends past the
    end of what has been delivered;
  * every granule's Huffman data ends exactly on part2_3_length (or all 576 lines were decoded and the rest is stuffing) -- with one wrong code
    length anywhere in the tables (scripts/mp3_tables.py) this fails within a few frames.
Steps (section numbers of the standard): header + side information (2.4.1.3-7), main data from the reservoir (2.4.2.7), scale factors, Huffman
pairs and quadruples (2.4.2.7, Annex B table B.7), requantisation (2.4.3.4.7.1), reordering of short blocks, mid-side, alias reduction (table B.9),
IMDCT with the four window shapes and overlap-add (2.4.3.4.10), frequency inversion, polyphase synthesis (Annex B figure A.2, window table B.3).
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mp3_tables as T  # noqa: E402

BITRATES = [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320]
RATES = [44100, 48000, 32000]
# scale-factor band widths, MPEG-1 (table B.8): long (22 bands) and short (13 bands) per sampling rate
SFB_LONG = {44100: [4, 4, 4, 4, 4, 4, 6, 6, 8, 8, 10, 12, 16, 20, 24, 28, 34, 42, 50, 54, 76, 158],
            48000: [4, 4, 4, 4, 4, 4, 6, 6, 6, 8, 10, 12, 16, 18, 22, 28, 34, 40, 46, 54, 54, 192],
            32000: [4, 4, 4, 4, 4, 4, 6, 6, 8, 10, 12, 16, 20, 24, 30, 38, 46, 56, 68, 84, 102, 26]}
SFB_SHORT = {44100: [4, 4, 4, 4, 6, 8, 10, 12, 14, 18, 22, 30, 56],
             48000: [4, 4, 4, 4, 6, 6, 10, 12, 14, 16, 20, 26, 66],
             32000: [4, 4, 4, 4, 6, 8, 12, 16, 20, 26, 34, 42, 12]}
PRETAB = [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 2, 0]
SLEN = [(0, 0), (0, 1), (0, 2), (0, 3), (3, 0), (1, 1), (1, 2), (1, 3), (2, 1), (2, 2), (2, 3), (3, 1), (3, 2), (3, 3), (4, 2), (4, 3)]
# table_select -> (tree, linbits)
HUFF_SEL = {0: (0, 0), 1: (1, 0), 2: (2, 0), 3: (3, 0), 5: (5, 0), 6: (6, 0), 7: (7, 0), 8: (8, 0), 9: (9, 0), 10: (10, 0), 11: (11, 0), 12: (12, 0),
            13: (13, 0), 15: (15, 0)}
for _i, _lb in enumerate([1, 2, 3, 4, 6, 8, 10, 13]):
    HUFF_SEL[16 + _i] = (16, _lb)
for _i, _lb in enumerate([4, 5, 6, 7, 8, 9, 11, 13]):
    HUFF_SEL[24 + _i] = (24, _lb)
QUAD_A_LEN = [1, 4, 4, 5, 4, 6, 5, 6, 4, 5, 5, 6, 5, 6, 6, 6]
QUAD_A_CODE = [1, 5, 4, 5, 6, 5, 4, 4, 7, 3, 6, 0, 7, 2, 3, 1]
ALIAS_C = [-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037]


def _trees():
    """per tree: {bit string: (x, y)} and the sorted distinct code lengths"""
    out = {}
    for t, (xs, lens, codes) in T.HUFF.items():
        d = {}
        for j, (n, c) in enumerate(zip(lens, codes)):
            d[format(c, "0%db" % n)] = (j // xs, j % xs)
        out[t] = (d, sorted(set(lens)))
    qa = {format(c, "0%db" % n): j for j, (n, c) in enumerate(zip(QUAD_A_LEN, QUAD_A_CODE))}
    return out, (qa, sorted(set(QUAD_A_LEN)))


TREES, QUAD_A = _trees()


def _imdct_tables():
    i36, k18 = np.arange(36)[:, None], np.arange(18)[None, :]
    c36 = np.cos(np.pi / 72.0 * (2 * i36 + 1 + 18) * (2 * k18 + 1))
    i12, k6 = np.arange(12)[:, None], np.arange(6)[None, :]
    c12 = np.cos(np.pi / 24.0 * (2 * i12 + 1 + 6) * (2 * k6 + 1))
    w = np.zeros((4, 36))
    i = np.arange(36)
    w[0] = np.sin(np.pi / 36.0 * (i + 0.5))
    w[1, :18] = np.sin(np.pi / 36.0 * (i[:18] + 0.5)); w[1, 18:24] = 1.0; w[1, 24:30] = np.sin(np.pi / 12.0 * (i[24:30] - 18 + 0.5))
    w[3, 6:12] = np.sin(np.pi / 12.0 * (i[6:12] - 6 + 0.5)); w[3, 12:18] = 1.0; w[3, 18:] = np.sin(np.pi / 36.0 * (i[18:] + 0.5))
    w12 = np.sin(np.pi / 12.0 * (np.arange(12) + 0.5))
    return c36, c12, w, w12


C36, C12, WIN36, WIN12 = _imdct_tables()
_c = np.array(ALIAS_C)
ALIAS_CS, ALIAS_CA = 1.0 / np.sqrt(1.0 + _c * _c), _c / np.sqrt(1.0 + _c * _c)
SYN_N = np.cos((16 + np.arange(64))[:, None] * (2 * np.arange(32)[None, :] + 1) * np.pi / 64.0)  # V[i] = sum_k N[i][k] S[k]
_d = np.zeros(512)
for _i in range(257):
    _d[_i] = T.DWIN[_i] / 65536.0
    if _i:
        _d[512 - _i] = -_d[_i] if (_i & 63) else _d[_i]
SYN_D = _d


def crc16_mpeg(bits: str, crc: int = 0xFFFF) -> int:
    """CRC-16 of the standard (polynomial 0x8005, initial 0xffff) over a bit string"""
    for ch in bits:
        top = (crc >> 15) & 1
        crc = (crc << 1) & 0xFFFF
        if top ^ (ch == "1"):
            crc ^= 0x8005
    return crc


class Granule:
    __slots__ = ("part2_3_length", "big_values", "global_gain", "scalefac_compress", "window_switching", "block_type", "mixed", "table_select",
                 "subblock_gain", "region0_count", "region1_count", "preflag", "scalefac_scale", "count1table_select")


def _side_info(bits: str, nch: int):
    p = 0

    def g(n):
        nonlocal p
        v = int(bits[p:p + n], 2) if n else 0
        p += n
        return v

    main_data_begin = g(9)
    g(5 if nch == 1 else 3)
    scfsi = [[g(1) for _ in range(4)] for _ in range(nch)]
    grs = [[None] * nch for _ in range(2)]
    for gr in range(2):
        for ch in range(nch):
            G = Granule()
            G.part2_3_length = g(12); G.big_values = g(9); G.global_gain = g(8); G.scalefac_compress = g(4); G.window_switching = g(1)
            if G.window_switching:
                G.block_type = g(2); G.mixed = g(1)
                G.table_select = [g(5), g(5), 0]
                G.subblock_gain = [g(3), g(3), g(3)]
                if G.block_type == 0:
                    raise ValueError("reserved block type")
                G.region0_count = 8 if (G.block_type == 2 and not G.mixed) else 7
                G.region1_count = 36
            else:
                G.block_type = 0; G.mixed = 0
                G.table_select = [g(5), g(5), g(5)]
                G.subblock_gain = [0, 0, 0]
                G.region0_count = g(4); G.region1_count = g(3)
            G.preflag = g(1); G.scalefac_scale = g(1); G.count1table_select = g(1)
            grs[gr][ch] = G
    return main_data_begin, scfsi, grs


def decode(path: str, max_seconds: float | None = None):
    """-> (pcm float64 [channels][n] in [-1, 1), sample rate, info dict)"""
    b = open(path, "rb").read()
    pos = 0
    if b[:3] == b"ID3":
        pos = 10 + ((b[6] & 0x7F) << 21 | (b[7] & 0x7F) << 14 | (b[8] & 0x7F) << 7 | (b[9] & 0x7F))
    reservoir = ""            # main-data bits delivered so far that a later frame may still point into (kept short)
    info = {"frames": 0, "resyncs": 0, "reservoir_underruns": 0, "granule_overruns": 0, "huffman_misfits": 0, "crc_checked": 0, "crc_bad": 0,
            "block_types": [0, 0, 0, 0], "mixed_blocks": 0, "ms_frames": 0, "max_main_data_begin": 0, "stuffing_bits": 0, "ancillary_bits": 0,
            "reservoir_overlaps": 0}
    delivered = used_end = 0  # main-data bits delivered by the frames so far / end of the last granule decoded, both as positions in that stream
    rate = nch = None
    subbands = []             # per frame: [nch][2][18][32] subband samples
    prev = None               # overlap [nch][32][18]
    sf_prev = None            # granule 0's long scale factors (scfsi)
    while pos + 4 <= len(b):
        h = int.from_bytes(b[pos:pos + 4], "big")
        if (h >> 21) != 0x7FF:
            if b[pos:pos + 3] == b"TAG" or b[pos:pos + 8].startswith(b"TAG"):
                break
            info["resyncs"] += 1
            break
        ver, layer, prot, bri, sri, pad, mode, mext = (h >> 19) & 3, (h >> 17) & 3, (h >> 16) & 1, (h >> 12) & 15, (h >> 10) & 3, (h >> 9) & 1, (h >> 6) & 3, (h >> 4) & 3
        if ver != 3 or layer != 1 or bri in (0, 15) or sri == 3:
            raise ValueError("not an MPEG-1 Layer III frame with a fixed bit rate at byte %d" % pos)
        fr, frame_nch = RATES[sri], (1 if mode == 3 else 2)
        if rate is None:
            rate, nch = fr, frame_nch
            prev = np.zeros((nch, 32, 18))
        assert (fr, frame_nch) == (rate, nch), "stream changes format"
        flen = 144 * BITRATES[bri] * 1000 // rate + pad
        if pos + flen > len(b):
            break
        si_len = 17 if nch == 1 else 32
        hp = pos + 4 + (0 if prot else 2)
        side = "".join(format(x, "08b") for x in b[hp:hp + si_len])
        if not prot:
            info["crc_checked"] += 1
            want = int.from_bytes(b[pos + 4:pos + 6], "big")
            got = crc16_mpeg(format(h & 0xFFFF, "016b") + side)
            info["crc_bad"] += int(want != got)
        main_data_begin, scfsi, grs = _side_info(side, nch)
        info["max_main_data_begin"] = max(info["max_main_data_begin"], main_data_begin)
        if mode == 1 and (mext & 1):
            raise NotImplementedError("intensity stereo")
        ms = mode == 1 and bool(mext & 2)
        info["ms_frames"] += int(ms)
        new_main = "".join(format(x, "08b") for x in b[hp + si_len:pos + flen])
        if main_data_begin * 8 > len(reservoir):
            # the first frames of a stream cut out of a longer one may point before its start: such a frame cannot be decoded (silence)
            info["reservoir_underruns"] += 1
            reservoir = (reservoir + new_main)[-8 * 4096:]
            delivered += len(new_main)
            used_end = delivered - len(new_main)
            subbands.append(np.zeros((nch, 2, 18, 32)))
            pos += flen
            info["frames"] += 1
            continue
        # absolute positions in the stream of main-data bits: this frame's granules start main_data_begin bytes before its own main data
        frame_start = delivered - main_data_begin * 8
        if frame_start < used_end:
            info["reservoir_overlaps"] += 1      # the frame points into bits the previous frame's granules already consumed
        else:
            info["ancillary_bits"] += frame_start - used_end
        data = reservoir[len(reservoir) - main_data_begin * 8:] + new_main
        delivered += len(new_main)
        p = 0
        out = np.zeros((nch, 2, 18, 32))
        sf_long_gr0 = [None] * nch
        for gr in range(2):
            xr = np.zeros((nch, 576))
            for ch in range(nch):
                G = grs[gr][ch]
                info["block_types"][G.block_type] += 1
                info["mixed_blocks"] += int(G.mixed)
                start = p
                end = start + G.part2_3_length
                if end > len(data):
                    info["granule_overruns"] += 1
                    raise ValueError("granule data extends past the delivered main data (frame %d)" % info["frames"])
                # ---- scale factors ----
                s1, s2 = SLEN[G.scalefac_compress]
                sf_l, sf_s = [0] * 22, [[0, 0, 0] for _ in range(13)]

                def g(n):
                    nonlocal p
                    v = int(data[p:p + n], 2) if n else 0
                    p += n
                    return v

                if G.window_switching and G.block_type == 2:
                    if G.mixed:
                        for sfb in range(8):
                            sf_l[sfb] = g(s1)
                        for sfb in range(3, 6):
                            for w in range(3):
                                sf_s[sfb][w] = g(s1)
                    else:
                        for sfb in range(6):
                            for w in range(3):
                                sf_s[sfb][w] = g(s1)
                    for sfb in range(6, 12):
                        for w in range(3):
                            sf_s[sfb][w] = g(s2)
                else:
                    for grp, (lo, hi, n) in enumerate(((0, 6, s1), (6, 11, s1), (11, 16, s2), (16, 21, s2))):
                        if gr == 1 and scfsi[ch][grp]:
                            sf_l[lo:hi] = sf_prev[ch][lo:hi]
                        else:
                            for sfb in range(lo, hi):
                                sf_l[sfb] = g(n)
                if gr == 0:
                    sf_long_gr0[ch] = list(sf_l)
                    sf_prev = sf_long_gr0
                # ---- Huffman ----
                long_w, short_w = SFB_LONG[rate], SFB_SHORT[rate]
                long_edges = np.concatenate([[0], np.cumsum(long_w)])
                if G.window_switching:
                    r1 = 36
                    r2 = 576
                else:
                    r1 = int(long_edges[G.region0_count + 1])
                    r2 = int(long_edges[min(G.region0_count + G.region1_count + 2, 22)])
                nbig = min(G.big_values * 2, 576)
                isv = [0] * 578
                i = 0
                while i < nbig:
                    tsel = G.table_select[0 if i < r1 else (1 if i < r2 else 2)]
                    if tsel not in HUFF_SEL:
                        raise ValueError("table %d does not exist" % tsel)
                    tree, linbits = HUFF_SEL[tsel]
                    if tree == 0:
                        i += 2
                        continue
                    d, lens = TREES[tree]
                    for n in lens:
                        v = d.get(data[p:p + n])
                        if v is not None:
                            p += n
                            break
                    else:
                        raise ValueError("no Huffman code matches (frame %d)" % info["frames"])
                    x, y = v
                    if linbits and x == 15:
                        x += int(data[p:p + linbits], 2); p += linbits
                    if x:
                        if data[p] == "1":
                            x = -x
                        p += 1
                    if linbits and y == 15:
                        y += int(data[p:p + linbits], 2); p += linbits
                    if y:
                        if data[p] == "1":
                            y = -y
                        p += 1
                    isv[i], isv[i + 1] = x, y
                    i += 2
                if p > end:
                    info["huffman_misfits"] += 1
                # count1 region: quadruples until the granule's bits are used up
                while p < end and i <= 572:
                    if G.count1table_select:
                        q = 15 - int(data[p:p + 4], 2); p += 4
                    else:
                        qa, qlens = QUAD_A
                        for n in qlens:
                            q = qa.get(data[p:p + n])
                            if q is not None:
                                p += n
                                break
                        else:
                            raise ValueError("no quadruple code matches")
                    for k, bit in enumerate((8, 4, 2, 1)):
                        if q & bit:
                            isv[i + k] = -1 if data[p] == "1" else 1
                            p += 1
                    i += 4
                if p > end:
                    # the last quadruple ran over the end: the standard says it is discarded
                    i -= 4
                    for k in range(4):
                        isv[i + k] = 0
                    info["huffman_misfits"] += 1   # a conforming encoder never produces this
                elif p < end:
                    if i < 576:
                        info["huffman_misfits"] += 1
                    info["stuffing_bits"] += end - p
                p = end
                # ---- requantise (and reorder short blocks) ----
                isa = np.array(isv[:576], dtype=np.float64)
                mag = np.sign(isa) * np.abs(isa) ** (4.0 / 3.0)
                mult = 1.0 if G.scalefac_scale else 0.5
                x = np.zeros(576)
                if G.window_switching and G.block_type == 2:
                    lines_long = 36 if G.mixed else 0
                    if G.mixed:
                        for sfb in range(8):
                            lo, hi = int(long_edges[sfb]), int(long_edges[sfb + 1])
                            e = 0.25 * (G.global_gain - 210) - mult * (sf_l[sfb] + G.preflag * PRETAB[sfb])
                            x[lo:hi] = mag[lo:hi] * 2.0 ** e
                    short_edges = np.concatenate([[0], np.cumsum(short_w)])
                    for sfb in range(3 if G.mixed else 0, 13):
                        s0, wdt = int(short_edges[sfb]), short_w[sfb]
                        for w in range(3):
                            e = 0.25 * (G.global_gain - 210 - 8 * G.subblock_gain[w]) - mult * sf_s[sfb][w]
                            src = mag[3 * s0 + w * wdt:3 * s0 + (w + 1) * wdt] * 2.0 ** e
                            x[3 * s0 + w + 3 * np.arange(wdt)] = src        # x[3 (s0 + j) + w] = decoded[3 s0 + w wdt + j]
                    assert lines_long in (0, 36)
                else:
                    for sfb in range(22):
                        lo, hi = int(long_edges[sfb]), int(long_edges[sfb + 1])
                        e = 0.25 * (G.global_gain - 210) - mult * (sf_l[sfb] + G.preflag * PRETAB[sfb])
                        x[lo:hi] = mag[lo:hi] * 2.0 ** e
                xr[ch] = x
            if ms:
                m, s = xr[0].copy(), xr[1].copy()
                xr[0], xr[1] = (m + s) / np.sqrt(2.0), (m - s) / np.sqrt(2.0)
            for ch in range(nch):
                G = grs[gr][ch]
                x = xr[ch]
                # ---- alias reduction ----
                if not (G.window_switching and G.block_type == 2 and not G.mixed):
                    nsb = 2 if (G.window_switching and G.block_type == 2) else 32
                    for sb in range(1, nsb):
                        lo = x[18 * sb - 1 - np.arange(8)].copy()
                        up = x[18 * sb + np.arange(8)].copy()
                        x[18 * sb - 1 - np.arange(8)] = lo * ALIAS_CS - up * ALIAS_CA
                        x[18 * sb + np.arange(8)] = up * ALIAS_CS + lo * ALIAS_CA
                # ---- IMDCT, windows, overlap-add, frequency inversion ----
                X = x.reshape(32, 18)
                y = np.zeros((32, 36))
                nlong = 32
                if G.window_switching and G.block_type == 2:
                    nlong = 2 if G.mixed else 0
                bt_long = 0 if (G.window_switching and G.block_type == 2) else G.block_type
                if nlong:
                    y[:nlong] = (X[:nlong] @ C36.T) * WIN36[bt_long][None, :]
                if nlong < 32:
                    Xs = X[nlong:].reshape(-1, 6, 3)        # [sb][k][window]: line 3 k + w
                    for w in range(3):
                        yw = (Xs[:, :, w] @ C12.T) * WIN12[None, :]
                        y[nlong:, 6 + 6 * w:18 + 6 * w] += yw
                res = y[:, :18] + prev[ch]
                prev[ch] = y[:, 18:]
                res[1::2, 1::2] *= -1.0
                out[ch, gr] = res.T      # [18 time slots][32 subbands]
        used_end = frame_start + p
        subbands.append(out)
        reservoir = data[-8 * 4096:] if len(data) > 8 * 4096 else data
        pos += flen
        info["frames"] += 1
        if max_seconds is not None and info["frames"] * 1152 >= max_seconds * rate:
            break
    # ---- polyphase synthesis, all time slots at once ----
    S = np.concatenate([sb.reshape(nch, 36, 32) for sb in subbands], axis=1)      # [nch][slots][32]
    nslots = S.shape[1]
    pcm = np.zeros((nch, nslots * 32))
    j = np.arange(32)
    for ch in range(nch):
        V = S[ch] @ SYN_N.T                                  # [slots][64]
        Vp = np.concatenate([np.zeros((16, 64)), V], axis=0)  # slot t lives at row t + 16
        acc = np.zeros((nslots, 32))
        t = np.arange(nslots) + 16
        for i in range(8):
            acc += SYN_D[j + 64 * i][None, :] * Vp[t - 2 * i][:, j]
            acc += SYN_D[j + 32 + 64 * i][None, :] * Vp[t - 2 * i - 1][:, 32 + j]
        pcm[ch] = acc.reshape(-1)
    info["seconds"] = pcm.shape[1] / float(rate)
    return pcm, rate, info


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/samples/10.5k_burst_sample.mp3"
    pcm, rate, info = decode(path, float(sys.argv[2]) if len(sys.argv) > 2 else None)
    print(info, "rate", rate, "shape", pcm.shape, "peak %.4f rms %.4f" % (np.abs(pcm).max(), np.sqrt(np.mean(pcm ** 2))))
    if len(sys.argv) > 3:
        np.clip(np.round(pcm[0] * 32768.0), -32768, 32767).astype("<i2").tofile(sys.argv[3])


if __name__ == "__main__":
    main()
