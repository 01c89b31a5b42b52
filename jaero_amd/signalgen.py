"""Synthetic Aero test signals (real 16-bit PCM at 48 kHz), as specified in SURVEY.md section 8(d).

The reference has no modulator; these follow the signal definitions its demodulators lock to
(probed in SURVEY.md 8(d) "Generator validity"):

* OQPSK 10.5 kbps: even bits on I, odd bits on Q, +-1 symbols at fb/2 symbols/s per arm, continuous-time
  root-raised-cosine alpha=1 pulse (the closed form of RootRaisedCosine::design, JAERO/DSP.h:316-338, evaluated
  at fractional sample offsets because Fs/(fb/2) = 9.142857 is not an integer), Q arm delayed half a symbol,
  x = I cos(2 pi fc n/Fs) - Q sin(2 pi fc n/Fs).
* MSK 600/1200 bps: CPFSK with modulation index 0.5 (+-fb/4), the hard decisions of the demodulator output
  equal the FSK data bits directly.

Everything here is numpy on the host (used by tests, the oracle legs and small benches).  `oqpsk_torch`
produces the same waveform family on a torch device for large resident bench inputs.
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 0x4A410000  # seed = 0x4A41_0000 + channel


def rrc_pulse(t: np.ndarray, T: float, alpha: float = 1.0) -> np.ndarray:
    """Continuous-time RRC impulse response sampled at offsets t (in samples); T = samples/symbol."""
    t = np.asarray(t, dtype=np.float64)
    out = np.empty_like(t)
    x = 4.0 * alpha * t / T
    centre = np.abs(t) < 1e-12
    sing = np.abs(1.0 - x * x) < 1e-10
    reg = ~(centre | sing)
    out[centre] = (4.0 * alpha + np.pi - np.pi * alpha) / (np.pi * np.sqrt(T))
    out[sing] = alpha * ((np.pi - 2.0) * np.cos(np.pi / (4.0 * alpha)) + (np.pi + 2.0) * np.sin(np.pi / (4.0 * alpha))) / (
        np.pi * np.sqrt(2.0 * T)
    )
    tr = t[reg]
    out[reg] = (
        4.0 * alpha / (np.pi * np.sqrt(T))
        * (np.cos((1.0 + alpha) * np.pi * tr / T) + T / (4.0 * alpha * tr) * np.sin((1.0 - alpha) * np.pi * tr / T))
        / (1.0 - (4.0 * alpha * tr / T) ** 2)
    )
    return out


def _shape(symbols: np.ndarray, n: np.ndarray, T: float, delay: float, span: int = 6) -> np.ndarray:
    """sum_k a_k h(n - delay - kT) over the 2*span+1 nearest symbols."""
    t = n - delay
    k0 = np.floor(t / T).astype(np.int64)
    acc = np.zeros(n.shape, dtype=np.float64)
    nsym = symbols.shape[0]
    for j in range(-span, span + 2):
        k = k0 + j
        ok = (k >= 0) & (k < nsym)
        a = np.where(ok, symbols[np.clip(k, 0, nsym - 1)], 0.0)
        acc += a * rrc_pulse(t - k * T, T, 1.0)
    return acc


def oqpsk(nsamples: int, *, fb: float = 10500.0, Fs: float = 48000.0, fc: float = 8000.0, ebno_db: float | None = 10.0,
          peak: float = 0.3, seed: int = SEED_BASE, bits: np.ndarray | None = None, start_sample: int = 0):
    """Returns (pcm int16[nsamples], bits uint8[...]) for a continuous 10.5k-style OQPSK channel."""
    rng = np.random.default_rng(seed)
    T = Fs / (fb / 2.0)
    nsym = int(np.ceil((start_sample + nsamples) / T)) + 16
    if bits is None:
        bits = rng.integers(0, 2, size=2 * nsym, dtype=np.uint8)
    else:
        # deterministic noise stream regardless of supplied bits
        rng.integers(0, 2, size=2 * nsym, dtype=np.uint8)
    a_i = 2.0 * bits[0::2][:nsym].astype(np.float64) - 1.0
    a_q = 2.0 * bits[1::2][:nsym].astype(np.float64) - 1.0
    n = np.arange(start_sample, start_sample + nsamples, dtype=np.float64)
    i_t = _shape(a_i, n, T, 0.0)
    q_t = _shape(a_q, n, T, T / 2.0)
    ph = 2.0 * np.pi * fc * n / Fs
    x = i_t * np.cos(ph) - q_t * np.sin(ph)
    p = float(np.mean(x * x)) if nsamples else 1.0
    if ebno_db is not None:
        sigma2 = p * Fs / (2.0 * fb * 10.0 ** (ebno_db / 10.0))
        x = x + rng.normal(0.0, np.sqrt(sigma2), size=nsamples)
    scale = peak / (3.0 * np.sqrt(p)) if p > 0 else 1.0  # 3-sigma-ish headroom, deterministic (no data-dependent max)
    pcm = np.clip(np.round(x * scale * 32768.0), -32768, 32767).astype(np.int16)
    return pcm, bits


def msk(nsamples: int, *, fb: float = 1200.0, Fs: float = 48000.0, fc: float = 1000.0, ebno_db: float | None = 10.0,
        peak: float = 0.3, seed: int = SEED_BASE, bits: np.ndarray | None = None):
    """Returns (pcm int16[nsamples], bits uint8[...]) for a continuous MSK (CPFSK h=0.5) channel."""
    rng = np.random.default_rng(seed)
    sps = Fs / fb
    nbits = int(np.ceil(nsamples / sps)) + 4
    if bits is None:
        bits = rng.integers(0, 2, size=nbits, dtype=np.uint8)
    else:
        rng.integers(0, 2, size=nbits, dtype=np.uint8)
    n = np.arange(nsamples, dtype=np.float64)
    k = np.minimum((n / sps).astype(np.int64), bits.shape[0] - 1)
    dev = (2.0 * bits[k].astype(np.float64) - 1.0) * (fb / 4.0)  # +-fb/4 Hz
    ph = 2.0 * np.pi * np.cumsum((fc + dev) / Fs)
    x = np.cos(ph)
    p = 0.5
    if ebno_db is not None:
        sigma2 = p * Fs / (2.0 * fb * 10.0 ** (ebno_db / 10.0))
        x = x + rng.normal(0.0, np.sqrt(sigma2), size=nsamples)
    pcm = np.clip(np.round(x * peak * 32768.0), -32768, 32767).astype(np.int16)
    return pcm, bits


def burst_oqpsk(nsamples: int, *, burst_starts, ndata_sym: int = 1500, fb: float = 10500.0, Fs: float = 48000.0, fc: float = 8000.0,
                ebno_db: float | None = 15.0, peak: float = 0.3, seed: int = SEED_BASE, noise_in_gaps: bool = True, data=None):
    """10.5 kbps burst OQPSK: per burst 128 symbols of constant (+1,+1) [carrier burst],
    128 symbols alternating +1,-1 on both arms [tones at fc +- fb/4: the "trident"], then `ndata_sym` random symbols
    per arm; same RRC alpha=1 pulse / half-symbol Q offset / passband law as `oqpsk`.  `burst_starts` are sample
    indices (rounded to the symbol grid).  Returns (pcm int16[nsamples], list of (start_sample, bits uint8[2*ndata_sym]))."""
    rng = np.random.default_rng(seed)
    T = Fs / (fb / 2.0)
    nsym = int(np.ceil(nsamples / T)) + 16
    a_i = np.zeros(nsym)
    a_q = np.zeros(nsym)
    bursts = []
    for st in burst_starts:
        k0 = int(round(st / T))
        nb = 256 + ndata_sym
        if k0 + nb > nsym:
            break
        bits = rng.integers(0, 2, size=2 * ndata_sym, dtype=np.uint8)
        if data is not None:  # caller's channel bits for this burst (e.g. unique word + an R/T packet), random fill behind them
            d = np.asarray(data[len(bursts)], dtype=np.uint8)
            bits[: len(d)] = d[: len(bits)]
        pre = np.concatenate([np.ones(128), np.where(np.arange(128) % 2 == 0, 1.0, -1.0)])
        a_i[k0:k0 + nb] = np.concatenate([pre, 2.0 * bits[0::2] - 1.0])
        a_q[k0:k0 + nb] = np.concatenate([pre, 2.0 * bits[1::2] - 1.0])
        bursts.append((int(round(k0 * T)), bits))
    n = np.arange(nsamples, dtype=np.float64)
    i_t = _shape(a_i, n, T, 0.0)
    q_t = _shape(a_q, n, T, T / 2.0)
    ph = 2.0 * np.pi * fc * n / Fs
    x = i_t * np.cos(ph) - q_t * np.sin(ph)
    p = 1.0 / T  # in-burst power of unit-symbol OQPSK with a unit-energy RRC pulse
    if ebno_db is not None:
        sigma2 = p * Fs / (2.0 * fb * 10.0 ** (ebno_db / 10.0))
        noise = rng.normal(0.0, np.sqrt(sigma2), size=nsamples)
        if not noise_in_gaps:
            noise = noise * ((np.abs(i_t) + np.abs(q_t)) > 0)
        x = x + noise
    scale = peak / (3.0 * np.sqrt(p))
    pcm = np.clip(np.round(x * scale * 32768.0), -32768, 32767).astype(np.int16)
    return pcm, bursts


def burst_msk(nsamples: int, *, burst_starts, ndata: int = 500, fb: float = 1200.0, Fs: float = 48000.0, fc: float = 1900.0,
              ncw: int = 120, npre: int = 100, ebno_db: float | None = 20.0, peak: float = 0.3, seed: int = SEED_BASE):
    """600 / 1200 bps burst "MSK" (R/T-channel style) as BurstMskDemodulator sees it (JAERO/burstmskdemodulator.cpp:444-569):
    `ncw` bit periods of unmodulated carrier (the base of the trident), `npre` bit periods of constant +1 symbols on both
    arms of an offset-QPSK signal with half-sine pulses of two bit periods (carrier plus lines at fc +- fb/2: the top of
    the trident), then `ndata` random bits (even bits on I, odd bits on Q, Q delayed one bit period).  The unmodified
    reference accepts these bursts and estimates fc within 2 Hz for ncw in 110..150.
    Returns (pcm int16[nsamples], list of (start_sample, bits uint8[ndata]))."""
    rng = np.random.default_rng(seed)
    sps = int(Fs / fb)
    T2 = 2 * sps
    x = np.zeros(nsamples, dtype=np.float64)
    pulse = np.sin(np.pi * np.arange(T2) / T2)
    bursts = []
    for st in burst_starts:
        ncws = ncw * sps
        L = ncws + (npre + ndata) * sps + 4 * sps
        bits = rng.integers(0, 2, size=ndata, dtype=np.uint8)
        a_i = np.concatenate([np.ones(npre // 2), 2.0 * bits[0::2] - 1.0])
        a_q = np.concatenate([np.ones(npre // 2), 2.0 * bits[1::2] - 1.0])
        i_t = np.zeros(L)
        q_t = np.zeros(L)
        for k in range(a_i.shape[0]):
            s0 = ncws + k * T2
            if s0 + T2 <= L:
                i_t[s0:s0 + T2] += a_i[k] * pulse
            if k < a_q.shape[0] and s0 + sps + T2 <= L:
                q_t[s0 + sps:s0 + sps + T2] += a_q[k] * pulse
        bb = i_t + 1j * q_t
        bb[:ncws] = (1.0 + 1.0j) / np.sqrt(2.0)
        n = np.arange(L, dtype=np.float64)
        sig = np.real(bb * np.exp(2j * np.pi * fc * (st + n) / Fs))
        m = max(0, min(L, nsamples - st))
        x[st:st + m] += sig[:m]
        bursts.append((int(st), bits))
    if ebno_db is not None:
        sigma2 = 0.5 * Fs / (2.0 * fb * 10.0 ** (ebno_db / 10.0))
        x = x + rng.normal(0.0, np.sqrt(sigma2), size=nsamples)
    pcm = np.clip(np.round(x * peak * 32768.0), -32768, 32767).astype(np.int16)
    return pcm, bursts


def channel_bank(kind: str, nch: int, nsamples: int, *, ebno_db: float | None = 10.0, seed0: int = SEED_BASE, **kw):
    """[nch, nsamples] int16 bank with per-channel carrier offsets as in SURVEY.md 8(d) configs 2/3.

    Returns (pcm, carriers, bits_list).  OQPSK: carrier 8000 + U(-100,100) Hz; MSK: 1000 + U(-50,50) Hz.
    """
    pcm = np.empty((nch, nsamples), dtype=np.int16)
    carriers = np.empty(nch, dtype=np.float64)
    bits_list = []
    for c in range(nch):
        r = np.random.default_rng(seed0 + c + 0x1000000)
        if kind == "oqpsk":
            fc = 8000.0 + r.uniform(-100.0, 100.0)
            p, b = oqpsk(nsamples, fc=fc, ebno_db=ebno_db, seed=seed0 + c, **kw)
        elif kind == "msk":
            fc = 1000.0 + r.uniform(-50.0, 50.0)
            p, b = msk(nsamples, fc=fc, ebno_db=ebno_db, seed=seed0 + c, **kw)
        else:
            raise ValueError(kind)
        pcm[c] = p
        carriers[c] = fc
        bits_list.append(b)
    return pcm, carriers, bits_list


# ----------------------------------------------------------------------------------------------------------------------
# torch (device-resident) generator for large banks: same waveform family as `oqpsk` above
# ----------------------------------------------------------------------------------------------------------------------
def _render_oqpsk_torch(a_i, a_q, carriers, nsamples: int, device, *, fb: float, Fs: float, sigma: float, scale: float, gen,
                        block: int = 2048, start: int = 0, tphase=None, nphase: int = 1):
    """Passband OQPSK (RRC alpha=1, Q arm delayed T/2) of the +-1/0 symbol arrays a_i, a_q [nch, nsym] -> int16 [nsamples, nch],
    samples start .. start+nsamples-1 of the stream.  tphase (int64 [nch], values < nphase) delays channel c's symbol clock by
    tphase[c]/nphase of TWO symbol periods (so both the symbol instants and the demodulator's odd/even symbol alternation are
    spread): real channels are not symbol-synchronous with each other."""
    import torch

    nch, nsym = a_i.shape
    T = Fs / (fb / 2.0)
    out = torch.empty((nsamples, nch), dtype=torch.int16, device=device)
    taus = torch.arange(nphase, device=device, dtype=torch.float64) * (2.0 * T / nphase)  # [nphase] timing offsets in samples

    def pulse(t):  # alpha = 1
        x = 4.0 * t / T
        den = 1.0 - x * x
        safe_t = torch.where(t.abs() < 1e-9, torch.ones_like(t), t)
        safe_den = torch.where(den.abs() < 1e-9, torch.ones_like(den), den)
        reg = 4.0 / (np.pi * np.sqrt(T)) * (torch.cos(2.0 * np.pi * safe_t / T)) / safe_den  # sin((1-alpha)..)=0 for alpha=1
        centre = (4.0 + np.pi - np.pi) / (np.pi * np.sqrt(T))
        sing = ((np.pi - 2.0) * np.cos(np.pi / 4.0) + (np.pi + 2.0) * np.sin(np.pi / 4.0)) / (np.pi * np.sqrt(2.0 * T))
        r = torch.where(den.abs() < 1e-9, torch.full_like(t, sing), reg)
        return torch.where(t.abs() < 1e-9, torch.full_like(t, centre), r)

    def shape(a, n, delay):
        if tphase is None:
            t = n - delay
            k0 = torch.floor(t / T).to(torch.int64)
            acc = torch.zeros((nch, n.shape[0]), dtype=torch.float32, device=device)
            for j in range(-6, 8):
                k = k0 + j
                ok = (k >= 0) & (k < nsym)
                kk = k.clamp(0, nsym - 1)
                h = pulse((t - k.to(torch.float64) * T)).to(torch.float32) * ok.to(torch.float32)
                acc += a[:, kk].to(torch.float32) * h[None, :]
            return acc
        # per-channel timing phase: the pulse samples are shared by the channels of one phase ([nphase, block] tables, gathered)
        t = n[None, :] - delay - taus[:, None]                       # [nphase, block]
        k0 = torch.floor(t / T).to(torch.int64)
        acc = torch.zeros((nch, n.shape[0]), dtype=torch.float32, device=device)
        for j in range(-6, 8):
            k = k0 + j
            ok = (k >= 0) & (k < nsym)
            h = pulse((t - k.to(torch.float64) * T)).to(torch.float32) * ok.to(torch.float32)  # [nphase, block]
            kk = k.clamp(0, nsym - 1)[tphase]                        # [nch, block]
            acc += torch.gather(a, 1, kk).to(torch.float32) * h[tphase]
        return acc

    for s in range(0, nsamples, block):
        e = min(nsamples, s + block)
        n = torch.arange(start + s, start + e, device=device, dtype=torch.float64)
        i_t = shape(a_i, n, 0.0)
        q_t = shape(a_q, n, T / 2.0)
        ph = (2.0 * np.pi / Fs) * carriers[:, None] * n[None, :]
        ph = torch.remainder(ph, 2.0 * np.pi).to(torch.float32)
        x = i_t * torch.cos(ph) - q_t * torch.sin(ph)
        if sigma > 0:
            x = x + torch.randn(x.shape, generator=gen, device=device, dtype=torch.float32) * sigma
        out[s:e] = torch.clamp(torch.round(x * scale), -32768, 32767).to(torch.int16).t()
    return out


class OqpskTorchStream:
    """A bank of continuous 10.5k-style OQPSK channels on a torch device, rendered piece by piece: render(start, n) -> int16 [n, nch]
    (frame-major).  Channel c: carrier fc_center + U(-fc_spread, fc_spread), random bits, AWGN at Eb/N0,
    and -- with nphase > 1 -- its own symbol-clock phase (one of `nphase` offsets over two symbol periods, drawn at random): channels
    of different satellites / transponders are not symbol-synchronous, and a demodulator bank's per-symbol work must not be timed on
    the special case where every channel of a wavefront reaches its symbol instant at the same sample."""

    def __init__(self, nch: int, total_samples: int, device, *, fb: float = 10500.0, Fs: float = 48000.0, fc_center: float = 8000.0,
                 fc_spread: float = 100.0, ebno_db: float | None = 10.0, peak: float = 0.3, seed: int = SEED_BASE, nphase: int = 32):
        import torch

        self.nch, self.total, self.device, self.fb, self.Fs, self.nphase = nch, total_samples, device, fb, Fs, nphase
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        T = Fs / (fb / 2.0)
        nsym = int(np.ceil(total_samples / T)) + 16
        # +-1 symbols of the two arms (bit 2k of channel c on I, bit 2k+1 on Q); the bits themselves are rebuilt on demand (bits_of)
        self.a_i = torch.randint(0, 2, (nch, nsym), generator=gen, device=device, dtype=torch.int8) * 2 - 1
        self.a_q = torch.randint(0, 2, (nch, nsym), generator=gen, device=device, dtype=torch.int8) * 2 - 1
        self.carriers = fc_center + (torch.rand(nch, generator=gen, device=device, dtype=torch.float64) * 2 - 1) * fc_spread
        self.tphase = torch.randint(0, nphase, (nch,), generator=gen, device=device) if nphase > 1 else None
        P = 1.0 / T  # signal power of unit-symbol OQPSK with a unit-energy RRC pulse
        self.sigma = float(np.sqrt(P * Fs / (2.0 * fb * 10.0 ** (ebno_db / 10.0)))) if ebno_db is not None else 0.0
        self.scale = peak / (3.0 * np.sqrt(P)) * 32768.0
        self.gen = gen

    def render(self, start: int, n: int, block: int = 1024):
        return _render_oqpsk_torch(self.a_i, self.a_q, self.carriers, n, self.device, fb=self.fb, Fs=self.Fs, sigma=self.sigma, scale=self.scale,
                                   gen=self.gen, block=block, start=start, tphase=self.tphase, nphase=self.nphase)

    def bits_of(self, c: int) -> np.ndarray:
        """Transmitted bits of channel c (uint8, even indices = I arm, odd = Q arm)."""
        i = ((self.a_i[c] + 1) // 2).to("cpu").numpy().astype(np.uint8)
        q = ((self.a_q[c] + 1) // 2).to("cpu").numpy().astype(np.uint8)
        out = np.empty(2 * len(i), np.uint8)
        out[0::2], out[1::2] = i, q
        return out

    def timing_offsets(self):
        """Per-channel symbol-clock delay in samples."""
        T = self.Fs / (self.fb / 2.0)
        return None if self.tphase is None else self.tphase.to("cpu").numpy() * (2.0 * T / self.nphase)


def oqpsk_torch(nch: int, nsamples: int, device, *, fb: float = 10500.0, Fs: float = 48000.0, fc_center: float = 8000.0,
                fc_spread: float = 100.0, ebno_db: float | None = 10.0, peak: float = 0.3, seed: int = SEED_BASE,
                block: int = 2048):
    """Returns (pcm int16 [nsamples, nch] frame-major on `device`, bits uint8 [nch, 2*nsym], carriers float64 [nch]).

    Channel c: carrier fc_center + U(-fc_spread, fc_spread), random bits, AWGN at Eb/N0, as SURVEY.md 8(d) config 3.
    """
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    T = Fs / (fb / 2.0)
    nsym = int(np.ceil(nsamples / T)) + 16
    bits = torch.randint(0, 2, (nch, 2 * nsym), generator=gen, device=device, dtype=torch.uint8)
    a_i = bits[:, 0::2].to(torch.float32) * 2 - 1
    a_q = bits[:, 1::2].to(torch.float32) * 2 - 1
    carriers = fc_center + (torch.rand(nch, generator=gen, device=device, dtype=torch.float64) * 2 - 1) * fc_spread
    P = 1.0 / T  # signal power of unit-symbol OQPSK with a unit-energy RRC pulse
    sigma = float(np.sqrt(P * Fs / (2.0 * fb * 10.0 ** (ebno_db / 10.0)))) if ebno_db is not None else 0.0
    scale = peak / (3.0 * np.sqrt(P)) * 32768.0
    out = _render_oqpsk_torch(a_i, a_q, carriers, nsamples, device, fb=fb, Fs=Fs, sigma=sigma, scale=scale, gen=gen, block=block)
    return out, bits, carriers


def burst_oqpsk_torch(nch: int, nsamples: int, device, *, period: int = 48000, ndata_sym: int = 3040, fb: float = 10500.0,
                      Fs: float = 48000.0, fc_center: float = 8000.0, fc_spread: float = 100.0, ebno_db: float | None = 15.0,
                      peak: float = 0.3, seed: int = SEED_BASE, block: int = 2048, max_offset_sym: int | None = None):
    """SURVEY.md 8(d) config 4 on a torch device: every channel sends one burst per `period` samples (128 symbols of
    carrier, 128 symbols of alternating preamble, `ndata_sym` random symbols per arm = <= 6080 bits), at a per-channel
    random offset inside the period (limited to `max_offset_sym` symbols when given: all channels' bursts then start within that
    window, so their acquisition events crowd into the same segments), noise in the gaps.
    Returns (pcm int16 [nsamples, nch], carriers, offsets)."""
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    T = Fs / (fb / 2.0)
    nsym = int(np.ceil(nsamples / T)) + 16
    psym = int(round(period / T))
    blen = 256 + ndata_sym
    assert blen < psym
    off = torch.randint(0, psym - blen if max_offset_sym is None else min(psym - blen, max_offset_sym), (nch,), generator=gen, device=device)
    k = torch.arange(nsym, device=device)[None, :]
    rel = torch.remainder(k - off[:, None], psym)  # symbol index inside the channel's period
    inb = rel < blen
    pre = torch.where(rel < 128, torch.ones_like(rel), torch.where(rel % 2 == 0, torch.ones_like(rel), -torch.ones_like(rel)))
    di = torch.randint(0, 2, (nch, nsym), generator=gen, device=device, dtype=torch.int8) * 2 - 1
    dq = torch.randint(0, 2, (nch, nsym), generator=gen, device=device, dtype=torch.int8) * 2 - 1
    a_i = torch.where(inb, torch.where(rel < 256, pre.to(torch.int8), di), torch.zeros_like(di))
    a_q = torch.where(inb, torch.where(rel < 256, pre.to(torch.int8), dq), torch.zeros_like(dq))
    carriers = fc_center + (torch.rand(nch, generator=gen, device=device, dtype=torch.float64) * 2 - 1) * fc_spread
    P = 1.0 / T
    sigma = float(np.sqrt(P * Fs / (2.0 * fb * 10.0 ** (ebno_db / 10.0)))) if ebno_db is not None else 0.0
    scale = peak / (3.0 * np.sqrt(P)) * 32768.0
    out = _render_oqpsk_torch(a_i, a_q, carriers, nsamples, device, fb=fb, Fs=Fs, sigma=sigma, scale=scale, gen=gen, block=block)
    return out, carriers, (off.to(torch.float64) * T)
