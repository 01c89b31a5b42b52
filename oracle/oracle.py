"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle.so (the plain-C restatement of the reference's demodulator path and of
libcorrect's Viterbi) plus a runner for oracle/_ref/jaero_ref (the UNMODIFIED reference sources built against
Qt 5.9.7 with shim headers).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; the product package jaero_amd/ never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_BIN = os.path.join(HERE, "_ref", "jaero_ref")

KIND_MSK, KIND_OQPSK, KIND_BURST_MSK, KIND_BURST_OQPSK = 0, 1, 2, 3


class Settings(C.Structure):
    _fields_ = [
        ("kind", C.c_int),
        ("coarsefreqest_fft_power", C.c_int),
        ("freq_center", C.c_double),
        ("lockingbw", C.c_double),
        ("fb", C.c_double),
        ("Fs", C.c_double),
        ("signalthreshold", C.c_double),
    ]


def oqpsk_settings(freq_center=8000.0, lockingbw=10500.0, fb=10500.0, Fs=48000.0, power=14, threshold=0.65):
    return Settings(KIND_OQPSK, power, freq_center, lockingbw, fb, Fs, threshold)


def msk_settings(freq_center=1000.0, lockingbw=1800.0, fb=1200.0, Fs=48000.0, power=13, threshold=0.5):
    return Settings(KIND_MSK, power, freq_center, lockingbw, fb, Fs, threshold)


def burst_oqpsk_settings(freq_center=8000.0, lockingbw=10500.0, fb=10500.0, Fs=48000.0, power=13, threshold=0.6):
    return Settings(KIND_BURST_OQPSK, power, freq_center, lockingbw, fb, Fs, threshold)


def burst_msk_settings(freq_center=1000.0, lockingbw=1800.0, fb=1200.0, Fs=48000.0, power=13, threshold=0.6):
    return Settings(KIND_BURST_MSK, power, freq_center, lockingbw, fb, Fs, threshold)


def build(force: bool = False) -> None:
    """Compile liboracle.so (and _ref when the reference tree is present)."""
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(
        os.path.getmtime(os.path.join(HERE, f)) for f in ("jaero_oracle.c", "jaero_oracle_burst.c", "viterbi_oracle.c", "jaero_oracle.h", "aerol_oracle.c", "aerol_oracle.h")
    ):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/JAERO") and os.path.exists("/opt/conda/bin/moc"):
        if force or not os.path.exists(REF_BIN):
            subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.jo_demod_create.restype = C.c_void_p
        L.jo_demod_create.argtypes = [C.POINTER(Settings)]
        L.jo_demod_destroy.argtypes = [C.c_void_p]
        L.jo_demod_set_settings.argtypes = [C.c_void_p, C.POINTER(Settings)]
        L.jo_demod_set_flags.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.jo_demod_set_dcd.argtypes = [C.c_void_p, C.c_int]
        L.jo_demod_center_freq_changed.argtypes = [C.c_void_p, C.c_double]
        L.jo_demod_write.restype = C.c_long
        L.jo_demod_write.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        for name in ("jo_demod_take_soft", "jo_demod_take_status", "jo_demod_take_symbols"):
            f = getattr(L, name)
            f.restype = C.c_long
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.jo_demod_capture_symbols.argtypes = [C.c_void_p, C.c_int]
        L.jo_demod_pending_soft.argtypes = [C.c_void_p]
        for name in ("jo_demod_get_mse", "jo_demod_get_freq_est", "jo_demod_get_freq_center"):
            f = getattr(L, name)
            f.restype = C.c_double
            f.argtypes = [C.c_void_p]
        L.jo_rrc_design.argtypes = [C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.jo_cis_table.argtypes = [C.c_void_p]
        L.jo_fft.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.jo_coarse_create.restype = C.c_void_p
        L.jo_coarse_create.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
        L.jo_coarse_destroy.argtypes = [C.c_void_p]
        L.jo_coarse_bigchange.argtypes = [C.c_void_p]
        L.jo_coarse_process.restype = C.c_double
        L.jo_coarse_process.argtypes = [C.c_void_p, C.c_void_p]
        L.jo_coarse_get_y.argtypes = [C.c_void_p, C.c_void_p]
        L.jo_codec_create.restype = C.c_void_p
        L.jo_codec_create.argtypes = [C.c_int]
        L.jo_codec_destroy.argtypes = [C.c_void_p]
        L.jo_codec_reset.argtypes = [C.c_void_p]
        L.jo_decode_continuous.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.jo_decode_soft.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.jo_encode_bits.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.jo_burst_create.restype = C.c_void_p
        L.jo_burst_create.argtypes = [C.POINTER(Settings)]
        L.jo_burst_destroy.argtypes = [C.c_void_p]
        L.jo_burst_set_flags.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.jo_burst_set_dcd.argtypes = [C.c_void_p, C.c_int]
        L.jo_burst_trace.argtypes = [C.c_void_p, C.c_int]
        L.jo_burst_capture_symbols.argtypes = [C.c_void_p, C.c_int]
        L.jo_burst_write.restype = C.c_long
        L.jo_burst_center_freq_changed.restype = None
        L.jo_burst_center_freq_changed.argtypes = [C.c_void_p, C.c_double]
        L.jo_burst_write.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        for name in ("jo_burst_take_soft", "jo_burst_take_events", "jo_burst_take_symbols"):
            f = getattr(L, name)
            f.restype = C.c_long
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.jo_burst_pending_soft.argtypes = [C.c_void_p]
        for name in ("jo_burst_get_mse", "jo_burst_get_freq_est"):
            f = getattr(L, name)
            f.restype = C.c_double
            f.argtypes = [C.c_void_p]
        L.jo_hilbert_kernel.argtypes = [C.c_int, C.c_void_p]
        L.jo_hilbert_create.restype = C.c_void_p
        L.jo_hilbert_create.argtypes = [C.c_int]
        L.jo_hilbert_destroy.argtypes = [C.c_void_p]
        L.jo_hilbert_latency.argtypes = [C.c_void_p]
        L.jo_hilbert_update.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        L.jo_aerol_create.restype = C.c_void_p
        L.jo_aerol_create.argtypes = [C.c_int]
        L.jo_aerol_create_burst.restype = C.c_void_p
        L.jo_aerol_create_burst.argtypes = [C.c_int]
        L.jo_aerol_take_packets.restype = C.c_long
        L.jo_aerol_take_packets.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.jo_aerol_destroy.argtypes = [C.c_void_p]
        L.jo_aerol_write.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.jo_aerol_take_sus.restype = C.c_long
        L.jo_aerol_take_sus.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.jo_aerol_take_events.restype = C.c_long
        L.jo_aerol_take_events.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.jo_aerol_take_voice.restype = C.c_long
        L.jo_aerol_take_voice.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.jo_aerol_dcd.argtypes = [C.c_void_p]
        L.jo_aerol_tick_dcd.argtypes = [C.c_void_p]
        L.jo_aerol_tick_dcd.restype = C.c_int
        _lib = L
    return _lib


class Demod:
    """One reference-semantics demodulator object (OqpskDemodulator or MskDemodulator)."""

    def __init__(self, settings: Settings, afc=False, sql=False, cpu_reduce=False, capture_symbols=False):
        self.L = lib()
        self.h = self.L.jo_demod_create(C.byref(settings))
        self.L.jo_demod_set_flags(self.h, int(afc), int(sql), int(cpu_reduce))
        self.L.jo_demod_set_dcd(self.h, 0)
        self.L.jo_demod_capture_symbols(self.h, int(capture_symbols))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.jo_demod_destroy(self.h)
            self.h = None

    def set_settings(self, s: Settings):
        self.L.jo_demod_set_settings(self.h, C.byref(s))

    def set_flags(self, afc, sql, cpu_reduce):
        self.L.jo_demod_set_flags(self.h, int(afc), int(sql), int(cpu_reduce))

    def set_dcd(self, dcd):
        self.L.jo_demod_set_dcd(self.h, int(dcd))

    def center_freq_changed(self, f):
        self.L.jo_demod_center_freq_changed(self.h, float(f))

    def write(self, pcm: np.ndarray):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        self.L.jo_demod_write(self.h, pcm.ctypes.data, pcm.shape[0])

    def take_soft(self) -> np.ndarray:
        out = []
        buf = np.empty(1 << 16, dtype=np.int16)
        while True:
            n = self.L.jo_demod_take_soft(self.h, buf.ctypes.data, buf.shape[0])
            out.append(buf[:n].copy())
            if n < buf.shape[0]:
                break
        return np.concatenate(out)

    def take_status(self) -> np.ndarray:
        out = []
        buf = np.empty((4096, 6), dtype=np.float64)
        while True:
            n = self.L.jo_demod_take_status(self.h, buf.ctypes.data, buf.shape[0])
            out.append(buf[:n].copy())
            if n < buf.shape[0]:
                break
        return np.concatenate(out)

    def take_symbols(self) -> np.ndarray:
        out = []
        buf = np.empty((1 << 14, 3), dtype=np.float64)
        while True:
            n = self.L.jo_demod_take_symbols(self.h, buf.ctypes.data, buf.shape[0])
            out.append(buf[:n].copy())
            if n < buf.shape[0]:
                break
        return np.concatenate(out)

    @property
    def mse(self):
        return self.L.jo_demod_get_mse(self.h)

    @property
    def freq_est(self):
        return self.L.jo_demod_get_freq_est(self.h)

    @property
    def freq_center(self):
        return self.L.jo_demod_get_freq_center(self.h)

    @property
    def pending(self):
        return self.L.jo_demod_pending_soft(self.h)


def run_demod(settings: Settings, pcm: np.ndarray, chunk=4096, afc=False, cpu_reduce=False,
              dcd_at: int = -1, capture_symbols=False, center_at: int = -1, center_hz: float = 0.0,
              set_at: int = -1, set_settings: Settings = None, sql=False, flags_events=(), dcd_off_at: int = -1):
    """Convenience: feed pcm in `chunk`-sample writes (an int, or the list of successive write sizes),
    return dict(soft, status[, symbols]).  set_at / set_settings: setSettings on the live object before the write that starts at or
    after that sample."""
    d = Demod(settings, afc=afc, sql=sql, cpu_reduce=cpu_reduce, capture_symbols=capture_symbols)
    flags_events = sorted(list(flags_events))  # [(sample, afc, sql, cpu_reduce)]: set_flags before the write that starts at or after that sample
    n = pcm.shape[0]
    s = 0
    sizes = None if isinstance(chunk, (int, np.integer)) else list(chunk)
    k = 0
    while s < n:
        if sizes is not None:
            chunk = sizes[k] if k < len(sizes) else n - s
            k += 1
        if dcd_at >= 0 and s >= dcd_at:
            d.set_dcd(1)
            dcd_at = -1
        if dcd_off_at >= 0 and s >= dcd_off_at:
            d.set_dcd(0)
            dcd_off_at = -1
        if center_at >= 0 and s >= center_at:
            d.center_freq_changed(center_hz)
            center_at = -1
        if set_at >= 0 and s >= set_at:
            d.set_settings(set_settings)
            set_at = -1
        while flags_events and s >= flags_events[0][0]:
            _, fa, fs_, fc = flags_events.pop(0)
            d.set_flags(fa, fs_, fc)
        m = min(chunk, n - s)
        d.write(pcm[s:s + m])
        s += m
    out = {"soft": d.take_soft(), "status": d.take_status(), "pending": d.pending, "mse": d.mse,
           "freq_est": d.freq_est, "freq_center": d.freq_center}
    if capture_symbols:
        out["symbols"] = d.take_symbols()
    return out


def _drain(fn, h, width, dtype, rows=1 << 14):
    out = []
    buf = np.empty((rows, width) if width > 1 else (rows,), dtype=dtype)
    while True:
        n = fn(h, buf.ctypes.data, rows)
        out.append(buf[:n].copy())
        if n < rows:
            break
    return np.concatenate(out)


class BurstDemod:
    """One reference-semantics burst demodulator object (BurstOqpskDemodulator or BurstMskDemodulator)."""

    def __init__(self, settings: Settings, afc=False, sql=False, cpu_reduce=False, capture_symbols=False, trace=False):
        self.L = lib()
        self.h = self.L.jo_burst_create(C.byref(settings))
        self.L.jo_burst_set_flags(self.h, int(afc), int(sql), int(cpu_reduce))
        self.L.jo_burst_capture_symbols(self.h, int(capture_symbols))
        self.L.jo_burst_trace(self.h, int(trace))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.jo_burst_destroy(self.h)
            self.h = None

    def set_dcd(self, dcd):
        self.L.jo_burst_set_dcd(self.h, int(dcd))

    def write(self, pcm: np.ndarray):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        self.L.jo_burst_write(self.h, pcm.ctypes.data, pcm.shape[0])

    def center_freq_changed(self, hz: float):
        self.L.jo_burst_center_freq_changed(self.h, float(hz))

    def set_settings(self, settings: Settings):
        """setSettings on the live object (burstoqpskdemodulator.cpp:202-277, burstmskdemodulator.cpp:150-325)."""
        self.L.jo_burst_set_settings.argtypes = [C.c_void_p, C.c_void_p]
        self.L.jo_burst_set_settings.restype = None
        self.L.jo_burst_set_settings(self.h, C.byref(settings))

    def take_soft(self):
        return _drain(self.L.jo_burst_take_soft, self.h, 1, np.int16, 1 << 16)

    def take_events(self):
        return _drain(self.L.jo_burst_take_events, self.h, 3, np.float64)

    def take_symbols(self):
        return _drain(self.L.jo_burst_take_symbols, self.h, 3, np.float64)

    @property
    def pending(self):
        return self.L.jo_burst_pending_soft(self.h)

    @property
    def mse(self):
        return self.L.jo_burst_get_mse(self.h)

    @property
    def freq_est(self):
        return self.L.jo_burst_get_freq_est(self.h)


def run_burst(settings: Settings, pcm: np.ndarray, chunk: int = 4096, afc=False, sql=False, capture_symbols=False, trace=False,
              center_at: int = -1, center_hz: float = 0.0, set_at=(), set_settings=None):
    """Feed pcm in `chunk`-sample writes (CenterFreqChangedSlot(center_hz) in front of the first write at or behind sample `center_at`;
    setSettings(set_settings) on the live object in front of the first write at or behind each sample of `set_at`);
    returns dict(soft, events[, symbols], pending, mse, freq_est)."""
    d = BurstDemod(settings, afc=afc, sql=sql, capture_symbols=capture_symbols, trace=trace)
    set_at = sorted([set_at] if np.isscalar(set_at) else list(set_at))
    for s in range(0, pcm.shape[0], chunk):
        if center_at >= 0 and s >= center_at:
            d.center_freq_changed(center_hz)
            center_at = -1
        while set_at and s >= set_at[0]:
            d.set_settings(set_settings if set_settings is not None else settings)
            set_at.pop(0)
        d.write(pcm[s:s + chunk])
    out = {"soft": d.take_soft(), "events": d.take_events(), "pending": d.pending, "mse": d.mse, "freq_est": d.freq_est}
    if capture_symbols:
        out["symbols"] = d.take_symbols()
    return out


def fastfir(x: np.ndarray, alpha=0.6, K=2048, nfft=4096, Fs=48000.0, fsym=5250.0) -> np.ndarray:
    """JFastFir with the RRC kernel of JAERO/tests/jfastfir_tests.cpp (restated overlap-add), complex128 in -> complex128 out."""
    L = lib()
    L.jo_fastfir_run.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]
    L.jo_fastfir_run.restype = None
    a = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.empty_like(a)
    L.jo_fastfir_run(a.ctypes.data, len(a), alpha, K, nfft, Fs, fsym, out.ctypes.data)
    return out


def hilbert_kernel(N=2048) -> np.ndarray:
    out = np.zeros(N, dtype=np.complex128)
    lib().jo_hilbert_kernel(N, out.ctypes.data)
    return out


def hilbert_stream(pcm: np.ndarray, chunk: int = 4096, N=2048):
    """QJHilbertFilter::update over a real int16 stream; returns (analytic complex128 [n], latency L)."""
    L = lib()
    h = L.jo_hilbert_create(N)
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    out = np.zeros(pcm.shape[0], dtype=np.complex128)
    for s in range(0, pcm.shape[0], chunk):
        m = min(chunk, pcm.shape[0] - s)
        L.jo_hilbert_update(h, pcm[s:].ctypes.data, m, out[s:].ctypes.data)
    lat = L.jo_hilbert_latency(h)
    L.jo_hilbert_destroy(h)
    return out, lat


class AeroL:
    """Continuous (P-channel) path of the reference's AeroL bit pipeline: soft bits -> signal units."""

    def __init__(self, fb: int, burst: bool = False):
        self.L = lib()
        self.h = self.L.jo_aerol_create_burst(int(fb)) if burst else self.L.jo_aerol_create(int(fb))
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            self.L.jo_aerol_destroy(self.h)
            self.h = None

    def write(self, soft: np.ndarray):
        soft = np.ascontiguousarray(soft, dtype=np.int16)
        self.L.jo_aerol_write(self.h, soft.ctypes.data, soft.shape[0])

    def take_sus(self) -> np.ndarray:
        """rows [frame, k, 12 bytes, crc_ok, frameinfo]"""
        return _drain(self.L.jo_aerol_take_sus, self.h, 16, np.int32)

    def take_events(self) -> np.ndarray:
        return _drain(self.L.jo_aerol_take_events, self.h, 3, np.int64)

    def take_packets(self) -> np.ndarray:
        """burst mode: rows [packet, chunk, 12 bytes, total bytes, type (1 R, 2 T)]"""
        return _drain(self.L.jo_aerol_take_packets, self.h, 16, np.int32)

    def take_voice(self):
        """C channel (fb 8400): (frame numbers uint32[n], voice bytes uint8[n, 300]) as handed to Voicesignal (aerol.cpp:2454-2481)"""
        rows = _drain(self.L.jo_aerol_take_voice, self.h, 304, np.uint8)
        rows = rows.reshape(-1, 304)
        return rows[:, :4].copy().view(np.uint32).reshape(-1), rows[:, 4:]

    @property
    def dcd(self):
        return self.L.jo_aerol_dcd(self.h)

    def tick_dcd(self) -> int:
        """AeroL::updateDCD (the reference's 1 s timer)."""
        return self.L.jo_aerol_tick_dcd(self.h)


def run_aerol(fb: int, soft: np.ndarray, group: int = 32):
    a = AeroL(fb)
    for s in range(0, soft.shape[0], group):
        a.write(soft[s:s + group])
    return {"sus": a.take_sus(), "events": a.take_events(), "dcd": a.dcd}


def demod_groups(soft: np.ndarray):
    """(start, end) of the groups a burst demodulator hands to processDemodulatedSoftBits (burstoqpskdemodulator.cpp:546-585): a
    start-of-burst marker (negative) is one entry, soft bits come in pairs, a group goes out once it holds >= 32 entries after a pair."""
    out, s, n = [], 0, len(soft)
    while s < n:
        e, cnt = s, 0
        while e < n:
            if soft[e] < 0:
                e += 1; cnt += 1
                continue
            step = 2 if e + 1 < n else 1
            e += step; cnt += step
            if cnt >= 32:
                break
        out.append((s, e))
        s = e
    return out


def run_aerol_burst(fb: int, soft: np.ndarray):
    a = AeroL(fb, burst=True)
    for s, e in demod_groups(soft):
        a.write(soft[s:e])
    return {"packets": a.take_packets(), "events": a.take_events(), "dcd": a.dcd}


def packets_from_rows(rows: np.ndarray):
    """[(type, bytes)] from take_packets rows."""
    out = []
    for pk in sorted(set(rows[:, 0].tolist())) if len(rows) else []:
        r = rows[rows[:, 0] == pk]
        r = r[np.argsort(r[:, 1])]
        data = bytes(int(v) for v in r[:, 2:14].reshape(-1))[: int(r[0, 14])]
        out.append((int(r[0, 15]), data))
    return out


def run_ref_aerol_burst(fb: int, soft: np.ndarray):
    """The unmodified AeroL in burst mode on demodulator-style groups: ([('R', 17 bytes) | ('T', 4 header bytes, n, [10-byte SUs])],
    number of ' Bad R/T Packet' lines, raw text)."""
    import re

    assert have_ref()
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "soft.s16"), os.path.join(td, "out.txt")
        np.ascontiguousarray(soft, dtype=np.int16).tofile(inp)
        env = dict(os.environ)
        env["QT_QPA_PLATFORM"] = "offscreen"
        subprocess.check_call([REF_BIN, "aerol", inp, outp, f"fb={fb}", "group=0", "burst=1"], env=env, stdout=subprocess.DEVNULL)
        txt = open(outp, "rb").read().decode("latin1")
    pk, bad = [], 0
    for line in txt.split("\n"):
        if "Bad R/T Packet" in line:
            bad += 1
        m = re.match(r"^((?: 0x[0-9A-F]{2}){17}) ", line)
        if m:
            pk.append(("R", bytes(int(x, 16) for x in m.group(1).split())))
            continue
        m = re.match(r"^ T Packet from AES: ([0-9A-F]{6}) to GES: ([0-9A-F]{2}) with (\d+) SUs", line)
        if m:
            pk.append(["T", bytes.fromhex(m.group(1) + m.group(2)), int(m.group(3)), []])
            continue
        m = re.match(r"^((?: 0x[0-9A-F]{2}){10})( |$)", line)
        if m and pk and isinstance(pk[-1], list):
            pk[-1][3].append(bytes(int(x, 16) for x in m.group(1).split()))
    return pk, bad, txt


def run_ref_aerol_c(soft: np.ndarray, group: int = 32):
    """The unmodified AeroL at fb = 8400 (DecodeC): (voice uint8[nframes, 300] from the Voicesignal(data, hex) emissions, list of the
    10 payload bytes of every sub-band signal unit it printed (crc ok, carrier detected, not a fill-in unit), DCD lines, raw text)."""
    import re

    assert have_ref()
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "soft.s16"), os.path.join(td, "out.txt")
        np.ascontiguousarray(soft, dtype=np.int16).tofile(inp)
        env = dict(os.environ)
        env["QT_QPA_PLATFORM"] = "offscreen"
        subprocess.check_call([REF_BIN, "aerol", inp, outp, "fb=8400", f"group={group}"], env=env, stdout=subprocess.DEVNULL)
        txt = open(outp, "rb").read().decode("latin1")
    voice, sus, dcd = [], [], []
    for line in txt.split("\n"):
        if line.startswith("#V "):
            voice.append(np.frombuffer(bytes.fromhex(line[3:].strip()), dtype=np.uint8))
        elif line.startswith("#DCD"):
            dcd.append(tuple(int(x) for x in line.split()[1:3]))
        else:
            m = re.match(r"^((?: 0x[0-9A-F]{2}){10}) ", line)
            if m:
                sus.append(bytes(int(x, 16) for x in m.group(1).split()))
    return (np.stack(voice) if voice else np.zeros((0, 300), np.uint8)), sus, dcd, txt


def run_ref_aerol(fb: int, soft: np.ndarray, group: int = 32):
    """The unmodified AeroL (oracle/_ref): returns (list of (k, 10 bytes, crc_ok) in output order, raw text)."""
    import re

    assert have_ref()
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "soft.s16"), os.path.join(td, "out.txt")
        np.ascontiguousarray(soft, dtype=np.int16).tofile(inp)
        env = dict(os.environ)
        env["QT_QPA_PLATFORM"] = "offscreen"
        subprocess.check_call([REF_BIN, "aerol", inp, outp, f"fb={fb}", f"group={group}"], env=env)
        txt = open(outp, "rb").read().decode("latin1")
    sus = []
    for line in txt.split("\n"):
        m = re.match(r"^(.)((?: 0x[0-9A-F]{2}){10})(.*)$", line)
        if m:
            k = ord(m.group(1)) - ord("0")
            b = bytes(int(x, 16) for x in m.group(2).split())
            sus.append((k, b, "Bad CRC" not in m.group(3)))
    return sus, txt


# ----------------------------------------------------------------------------------------------- _ref runner
def have_ref() -> bool:
    return os.path.exists(REF_BIN) and os.access(REF_BIN, os.X_OK)


def run_ref(kind: str, pcm: np.ndarray, **kv):
    """Run the unmodified reference demodulator (one process = one channel).  kv -> key=value driver options."""
    assert have_ref(), "oracle/_ref/jaero_ref not built (needs /root/reference + /opt/conda Qt; see oracle/Makefile)"
    with tempfile.TemporaryDirectory() as td:
        inp = os.path.join(td, "in.s16")
        np.ascontiguousarray(pcm, dtype=np.int16).tofile(inp)
        outp = os.path.join(td, "out")
        args = [REF_BIN, kind, inp, outp] + [f"{k}={v}" for k, v in kv.items()]
        subprocess.check_call(args)
        soft = np.fromfile(outp + ".soft", dtype=np.int16)
        if kind.startswith("burst"):
            events = np.fromfile(outp + ".events", dtype=np.float64).reshape(-1, 3)
            return {"soft": soft, "events": events}
        status = np.fromfile(outp + ".status", dtype=np.float64).reshape(-1, 6)
    return {"soft": soft, "status": status}


def time_ref(kind: str, pcm_path: str, **kv) -> float:
    out = subprocess.check_output([REF_BIN, "time", kind, pcm_path] + [f"{k}={v}" for k, v in kv.items()])
    return float(out.split()[0])


def ref_tool(mode: str, data: np.ndarray, out_dtype, **kv) -> np.ndarray:
    assert have_ref()
    with tempfile.TemporaryDirectory() as td:
        inp = os.path.join(td, "in.bin")
        np.ascontiguousarray(data).tofile(inp)
        outp = os.path.join(td, "out.bin")
        subprocess.check_call([REF_BIN, mode, inp, outp] + [f"{k}={v}" for k, v in kv.items()])
        return np.fromfile(outp, dtype=out_dtype)


# ----------------------------------------------------------------------------------------------- Viterbi
class Codec:
    """JConvolutionalCodec restatement (K=7 r=1/2 {109,79}, padding 24 as AeroL uses)."""

    def __init__(self, paddinglength=24):
        self.L = lib()
        self.h = self.L.jo_codec_create(paddinglength)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.jo_codec_destroy(self.h)
            self.h = None

    def reset(self):
        self.L.jo_codec_reset(self.h)

    def decode_continuous(self, soft: np.ndarray) -> np.ndarray:
        soft = np.ascontiguousarray(soft, dtype=np.uint8)
        out = np.zeros(soft.shape[0] // 2 + 8, dtype=np.uint8)
        n = self.L.jo_decode_continuous(self.h, soft.ctypes.data, soft.shape[0], out.ctypes.data)
        return out[:n]

    def decode_soft(self, soft: np.ndarray) -> np.ndarray:
        soft = np.ascontiguousarray(soft, dtype=np.uint8)
        out = np.zeros(soft.shape[0] // 2 + 8, dtype=np.uint8)
        n = self.L.jo_decode_soft(self.h, soft.ctypes.data, soft.shape[0], out.ctypes.data)
        return out[:n]


def encode_bits(msg_bytes: np.ndarray) -> np.ndarray:
    """libcorrect-style encode of whole bytes; returns coded bits (0/1), 2*(8*len+8) of them."""
    msg_bytes = np.ascontiguousarray(msg_bytes, dtype=np.uint8)
    out = np.zeros(2 * (8 * msg_bytes.shape[0] + 8) + 16, dtype=np.uint8)
    n = lib().jo_encode_bits(msg_bytes.ctypes.data, msg_bytes.shape[0], out.ctypes.data)
    return out[:n]
