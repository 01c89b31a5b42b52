"""Synthetic Aero-L P-channel frames (the inverse of the reference's AeroL::Decode continuous path), for tests and benches.

The reference only decodes; this builds what it expects to see, from the decoder's own constants:
  * signal units: 12 bytes = 10 payload + CRC-16 (AeroLcrc16::calcusingbytes, poly 0x8408 reflected, init 0xFFFF, inverted; low
    byte first) -- JAERO/aerol.h:283-392, JAERO/aerol.cpp:1586-1589
  * information field bits LSB first per byte (aerol.cpp:1566-1578), XOR the 15-bit LFSR sequence restarted every frame
    (AeroLScrambler, aerol.h:394-440)
  * K=7 rate-1/2 convolutional code {109, 79}, newest bit in the LSB of the shift register, polynomial 0 sent first
    (libcorrect as JConvolutionalCodec uses it, aerol.cpp:936-940), run continuously over the frames
  * 64 x N block interleaver with row permutation (i*27)%64, N = 6 / 9 / 78 for 600 / 1200 / 10500 bps
    (AeroLInterleaver, aerol.cpp:523-625, 1013-1052)
  * frame = unique word 0xE15AE893 (on each of the I and Q arms for 10500: 64 channel bits) + 16 header bits
    (format id, super-frame marker, frame counter twice; aerol.cpp:1292-1296) + 178 dummy bits (10500 only) + coded bits:
    1200 channel bits (600/1200 bps) or 5250 (10500 bps).
The decoder's Viterbi (6 bits) + delay line make the signal units of frame f appear while frame f+1 (600/1200) or f+2 (10500) is
being received (aerol.cpp:1557-1560).
"""
from __future__ import annotations

import numpy as np

UW = 0xE15AE893
POLYS = (109, 79)


def crc16(data: bytes) -> int:
    crc = 0xFFFF
    for byte in data:
        for k in range(8):
            bit = (byte >> k) & 1
            c = crc & 1
            crc >>= 1
            if c ^ bit:
                crc ^= 0x8408
    return (~crc) & 0xFFFF


def make_su(payload10: bytes) -> bytes:
    assert len(payload10) == 10
    c = crc16(payload10)
    return bytes(payload10) + bytes([c & 0xFF, c >> 8])


def scrambler_sequence(n: int = 5000) -> np.ndarray:
    state = [1, 1, 0, 1, 0, 0, 1, 0, 1, 0, 1, 1, 0, 0, 1]
    out = np.zeros(n, dtype=np.uint8)
    for a in range(n):
        v = state[0] ^ state[14]
        out[a] = v
        state = [v] + state[:-1]
    return out


def conv_encode(bits: np.ndarray) -> np.ndarray:
    """Continuous K=7 r=1/2 encode of a 0/1 array (no tail); returns 2*len coded bits."""
    par = np.array([bin(i).count("1") & 1 for i in range(128)], dtype=np.uint8)
    out = np.zeros(2 * len(bits), dtype=np.uint8)
    sr = 0
    for i, b in enumerate(bits):
        sr = ((sr << 1) | int(b)) & 127
        out[2 * i] = par[sr & POLYS[0]]
        out[2 * i + 1] = par[sr & POLYS[1]]
    return out


def interleave(coded: np.ndarray, ncols: int) -> np.ndarray:
    """One 64 x ncols block: the decoder reads deinterleaved[j*64+i] = received[((i*27)%64)*ncols + j]."""
    assert len(coded) == 64 * ncols
    rx = np.zeros_like(coded)
    i = np.arange(64)
    for j in range(ncols):
        rx[((i * 27) % 64) * ncols + j] = coded[j * 64 + i]
    return rx


def geometry(fb: int):
    if fb == 10500:
        return dict(ncols=78, header=16, dummy=178, nbits=4992, uw=64, blocks=1, oqpsk=True, delay_frames=2)
    if fb in (600, 1200):
        n = 6 if fb == 600 else 9
        return dict(ncols=n, header=16, dummy=0, nbits=1152, uw=32, blocks=1152 // (64 * n), oqpsk=False, delay_frames=1)
    raise ValueError(fb)


def p_channel_bits(frames_payload, fb: int = 10500, *, first_counter: int = 0, invert_i: bool = False, invert_q: bool = False):
    """frames_payload: list (per frame) of lists of 10-byte payloads (26 signal units per frame at 10500, 6 at 600/1200;
    missing ones are filled with all-zero units, which the decoder accepts).  Returns (channel bits uint8, frame length)."""
    g = geometry(fb)
    nsu = g["nbits"] // 2 // 8 // 12
    scr = scrambler_sequence()
    msg = []
    for pay in frames_payload:
        sus = [make_su(p) for p in pay] + [bytes(12)] * (nsu - len(pay))
        info = b"".join(sus[:nsu])
        b = np.unpackbits(np.frombuffer(info, dtype=np.uint8), bitorder="little")
        msg.append(b ^ scr[: len(b)])
    coded = conv_encode(np.concatenate(msg))
    uwbits = np.array([(UW >> (31 - k)) & 1 for k in range(32)], dtype=np.uint8)
    out = []
    per_frame = g["nbits"]
    blk = 64 * g["ncols"]
    for f in range(len(frames_payload)):
        fc = (first_counter + f) & 15
        hdr_val = (1 << 12) | (0 << 8) | (fc << 4) | fc  # format id 1, super-frame marker 0, frame counter twice
        hdr = np.array([(hdr_val >> (15 - k)) & 1 for k in range(16)], dtype=np.uint8)
        body = coded[f * per_frame:(f + 1) * per_frame]
        body = np.concatenate([interleave(body[k * blk:(k + 1) * blk], g["ncols"]) for k in range(per_frame // blk)])
        if g["oqpsk"]:
            uw = np.repeat(uwbits, 2)  # the same 32 bits on the imag (even positions) and real (odd positions) arms
            fr = np.concatenate([uw, hdr, np.zeros(g["dummy"], np.uint8), body])
        else:
            fr = np.concatenate([uwbits, hdr, body])
        out.append(fr)
    bits = np.concatenate(out)
    if g["oqpsk"] and (invert_i or invert_q):
        bits = bits.copy()
        if invert_i:
            bits[0::2] ^= 1
        if invert_q:
            bits[1::2] ^= 1
    return bits, len(out[0])


def to_soft(bits: np.ndarray, *, sigma: float = 0.0, seed: int = 0) -> np.ndarray:
    """Channel bits -> soft bits as the demodulators emit them (int16 0..255, 128 = erasure): 0 -> ~53, 1 -> ~203 plus noise."""
    rng = np.random.default_rng(seed)
    x = (bits.astype(np.float64) * 2 - 1) * 75.0 + 128.0
    if sigma > 0:
        x = x + rng.normal(0.0, sigma, size=x.shape)
    return np.clip(np.round(x), 0, 255).astype(np.int16)


def random_payloads(nframes: int, fb: int, seed: int = 0):
    g = geometry(fb)
    nsu = g["nbits"] // 2 // 8 // 12
    rng = np.random.default_rng(seed)
    return [[bytes(rng.integers(0, 256, size=10, dtype=np.uint8)) for _ in range(nsu)] for _ in range(nframes)]


# ------------------------------------------------------------------------------------------------- R / T channel bursts (10500 bps)
def crc16_bits(bits: np.ndarray) -> np.ndarray:
    """The 16 CRC bits AeroLcrc16::calcusingbitsandcheck expects after `bits` (aerol.h:287-315): ~crc, least significant bit first."""
    crc = 0xFFFF
    for b in bits:
        c = crc & 1
        crc >>= 1
        if c ^ int(b):
            crc ^= 0x8408
    crc = (~crc) & 0xFFFF
    return np.array([(crc >> k) & 1 for k in range(16)], dtype=np.uint8)


def _bytes_to_bits(data: bytes) -> np.ndarray:
    return np.unpackbits(np.frombuffer(data, dtype=np.uint8), bitorder="little")


def rt_packet_bits(kind: str, payload) -> np.ndarray:
    """Channel bits of one R packet (`payload`: 17 bytes) or T packet (`payload`: (4-byte header, [10-byte SUs], at least two)) as
    RTChannelDeleaveFECScram::update expects them (aerol.h:785-873): [fields + CRC-16 each] -> scrambled -> K=7 r=1/2 from the zero
    state, flushed -> 64 x cols block interleaver with cols = 5 (R) or 5 + 3 (n - 1) (T with n signal units)."""
    if kind == "R":
        assert len(payload) == 17
        b = _bytes_to_bits(bytes(payload))
        info = np.concatenate([b, crc16_bits(b)])
        cols = 5
    else:
        hdr, sus = payload
        assert len(hdr) == 4 and len(sus) >= 2 and all(len(x) == 10 for x in sus)
        parts = []
        for field in [bytes(hdr)] + [bytes(x) for x in sus]:
            b = _bytes_to_bits(field)
            parts += [b, crc16_bits(b)]
        info = np.concatenate(parts)
        cols = 5 + 3 * (len(sus) - 1)
    ndec = 32 * cols
    assert len(info) + 6 <= ndec
    msg = np.zeros(ndec, dtype=np.uint8)
    msg[: len(info)] = info ^ scrambler_sequence(len(info))
    coded = conv_encode(msg)
    return interleave(coded, cols)


def rt_burst_stream(packets, *, gap: int = 12000, lead: int = 80, sigma: float = 20.0, seed: int = 0, invert_i=False, invert_q=False):
    """Soft-bit stream of a burst demodulator carrying the given packets [(kind, payload), ..]: noise, then per burst the start-of-burst
    marker (-1), `lead` soft bits of noise, the unique word on both arms, the packet, and `gap` soft bits of noise."""
    rng = np.random.default_rng(seed)
    noise = lambda n: np.clip(np.round(128 + rng.normal(0, 40, n)), 0, 255).astype(np.int16)
    uwbits = np.repeat(np.array([(UW >> (31 - k)) & 1 for k in range(32)], dtype=np.uint8), 2)
    out = [noise(64)]
    for kind, payload in packets:
        bits = np.concatenate([uwbits, rt_packet_bits(kind, payload)])
        if invert_i:
            bits[0::2] ^= 1
        if invert_q:
            bits[1::2] ^= 1
        ld = lead + (lead & 1)
        out += [np.array([-1], dtype=np.int16), noise(ld), to_soft(bits, sigma=sigma, seed=int(rng.integers(1 << 30))), noise(gap + (gap & 1))]
    return np.concatenate(out)


def interleave_msk(coded: np.ndarray) -> np.ndarray:
    """The inverse of AeroLInterleaver::deinterleaveMSK_ba (aerol.cpp:671-711): the first five 64-bit columns form one 64 x 5 block,
    every following three columns a 64 x 3 block of their own."""
    assert len(coded) % 64 == 0 and (len(coded) // 64 - 5) % 3 == 0
    out = np.zeros_like(coded)
    i = np.arange(64)
    k = 0
    for j in range(5):
        out[((i * 27) % 64) * 5 + j] = coded[k + i]
        k += 64
    proc = 5
    while k < len(coded):
        for j in range(3):
            out[64 * proc + ((i * 27) % 64) * 3 + j] = coded[k + i]
            k += 64
        proc += 3
    return out


def rt_packet_bits_msk(kind: str, payload) -> np.ndarray:
    """600 / 1200 bps R or T packet for RTChannelDeleaveFECScram::updateMSK (aerol.h:631-782).  A T packet is found through the count
    its SECOND signal unit carries in the low six bits of its first byte: targetSUSize = 2 + count (halved + 1 from 16 up), and the
    block is decoded at (targetSUSize + 1) * 3 + 2 columns, which hold targetSUSize + 1 units.  `payload` for T: (4-byte header,
    list of 10-byte units, at least 4); the count field of unit 1 is overwritten accordingly."""
    if kind == "R":
        b = _bytes_to_bits(bytes(payload))
        info = np.concatenate([b, crc16_bits(b)])
        cols = 5
    else:
        hdr, sus = payload
        sus = [bytearray(x) for x in sus]
        T = len(sus) - 1
        assert 3 <= T < 16
        sus[1][0] = (sus[1][0] & 0xC0) | (T - 2)
        parts = []
        for field in [bytes(hdr)] + [bytes(x) for x in sus]:
            b = _bytes_to_bits(field)
            parts += [b, crc16_bits(b)]
        info = np.concatenate(parts)
        cols = (T + 1) * 3 + 2
    ndec = 32 * cols
    assert len(info) + 6 <= ndec
    msg = np.zeros(ndec, dtype=np.uint8)
    msg[: len(info)] = info ^ scrambler_sequence(len(info))
    return interleave_msk(conv_encode(msg))


def rt_burst_stream_msk(packets, *, gap: int = 4200, lead: int = 80, sigma: float = 20.0, seed: int = 0, invert=False):
    """As rt_burst_stream, for the 600 / 1200 bps burst demodulators: one bit stream, the unique word once."""
    rng = np.random.default_rng(seed)
    noise = lambda n: np.clip(np.round(128 + rng.normal(0, 40, n)), 0, 255).astype(np.int16)
    uwbits = np.array([(UW >> (31 - k)) & 1 for k in range(32)], dtype=np.uint8)
    out = [noise(64)]
    for kind, payload in packets:
        bits = np.concatenate([uwbits, rt_packet_bits_msk(kind, payload)])
        if invert:
            bits = bits ^ 1
        out += [np.array([-1], dtype=np.int16), noise(lead + (lead & 1)), to_soft(bits, sigma=sigma, seed=int(rng.integers(1 << 30))), noise(gap + (gap & 1))]
    return np.concatenate(out)


# ------------------------------------------------------------------------------------------------ 8400 bps C channel
# The inverse of AeroL::DecodeC (JAERO/aerol.cpp:2187-2502, setSettings case 8400 :1033-1043):
#   frame = 104 unique-word channel bits (52 on each arm, real arm first: OQPSKPreambleDetectorAndAmbiguityCorrection accepts either
#   of its two words, or its complement, on either arm, tolerance 6) + 4096 channel bits = 4200 bits = 0.5 s;
#   the 4096 = 16 interleaver blocks of 64 x 4; deinterleaved they are the rate-3/4 punctured stream (every 4th coded bit of the
#   K=7 rate-1/2 code is not sent; the decoder ignores the last channel bit: 4095 -> 5460 soft symbols -> 2730 bits, of which it
#   keeps 2714);  the Viterbi's 6 bits + the 2708-bit delay line line the frames up one frame later, so bits 2708..2713 of a
#   frame come out of the 16 dropped positions' successors: the six bits wanted there are sent at positions 2724..2729;
#   the 2714 bits are scrambled (sequence restarted at every unique word) and hold 25 primary fields of 1 + 96 (voice) bits and
#   24 sub-band fields of 12 bits: three 12-byte signal units (10 + CRC-16) per frame.
C_UW1, C_UW2 = 216866263330005, 3012071630031408


def c_channel_bits(frames, *, invert_real: bool = False, invert_imag: bool = False, uw=(C_UW1, C_UW2)):
    """frames: list of (voice uint8[300], [three 10-byte payloads]) -> channel bits uint8 (4200 per frame).  The decoder hands out
    frame f while it receives frame f + 1."""
    scr = scrambler_sequence()
    info = []
    for voice, pays in frames:
        vb = np.unpackbits(np.asarray(voice, dtype=np.uint8), bitorder="little")  # 2400 bits, LSB first per byte (aerol.cpp:2461-2469)
        sub = np.unpackbits(np.frombuffer(b"".join(make_su(bytes(p)) for p in pays), dtype=np.uint8), bitorder="little")  # 288 bits
        d = np.zeros(2714, dtype=np.uint8)
        for y in range(25):
            d[y * 109 + 1: y * 109 + 97] = vb[y * 96:(y + 1) * 96]
            if y < 24:
                d[y * 109 + 97: y * 109 + 109] = sub[y * 12:(y + 1) * 12]
        d ^= scr[:2714]
        info.append(np.concatenate([d[:2708], np.zeros(16, np.uint8), d[2708:]]))
    coded = conv_encode(np.concatenate(info))
    keep = (np.arange(5460) % 4) != 3
    out = []
    for f in range(len(frames)):
        tx = coded[f * 5460:(f + 1) * 5460][keep]          # 4095 channel bits
        tx = np.concatenate([tx, np.zeros(1, np.uint8)])      # the 4096th is never looked at
        body = np.concatenate([interleave(tx[k * 256:(k + 1) * 256], 4) for k in range(16)])
        wr = np.array([(uw[0] >> (51 - k)) & 1 for k in range(52)], dtype=np.uint8)
        wi = np.array([(uw[1] >> (51 - k)) & 1 for k in range(52)], dtype=np.uint8)
        u = np.empty(104, dtype=np.uint8)
        u[0::2], u[1::2] = wr, wi
        out.append(np.concatenate([u, body]))
    bits = np.concatenate(out)
    if invert_real:
        bits[0::2] ^= 1  # the stream starts on the real arm (realimag toggles before it is tested, aerol.cpp:2208)
    if invert_imag:
        bits[1::2] ^= 1
    return bits


def c_channel_case(seed: int, nframes: int, sigma: float, inv=(False, False), lead: int = 74, types=(0x22, 0x30, 0x60, 0x01)):
    """A test stream: ([(voice uint8[300], [three 10-byte payloads])] per frame, soft bits int16).  Message types that DecodeC prints
    (and a fill-in unit now and then); `lead` random bits in front so that the unique word does not start the stream."""
    rng = np.random.default_rng(seed)
    frames = [(rng.integers(0, 256, 300, dtype=np.uint8),
               [bytes([types[int(rng.integers(0, len(types)))]] + list(rng.integers(0, 256, 9, dtype=np.uint8))) for _ in range(3)])
              for _ in range(nframes)]
    bits = c_channel_bits(frames, invert_real=inv[0], invert_imag=inv[1])
    soft = to_soft(np.concatenate([rng.integers(0, 2, lead, dtype=np.uint8), bits, np.zeros(300, np.uint8)]), sigma=sigma, seed=seed + 1)
    return frames, soft

