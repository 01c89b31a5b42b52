"""Host-side mirror of the reference's demodulator surface on top of the C ABI.

`DemodulatorBank` is the batched object (one bank = N channels of one kind on one GPU).  `OqpskDemodulator` and
`MskDemodulator` are single-channel views with the reference's own method names (setSettings, setAFC, setSQL,
setCPUReduce, DCDstatSlot, CenterFreqChangedSlot, start/stop, writeData) and its signal
`processDemodulatedSoftBits` as a Python callback, so tests read like the reference's call sites
(JAERO/mainwindow.cpp:198-202,234-237,816-899).  All arithmetic happens in libjaero_hip.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np

from . import capi


@dataclass
class OqpskSettings:
    """OqpskDemodulator::Settings defaults (JAERO/oqpskdemodulator.h:20-39)."""

    coarsefreqest_fft_power: int = 14
    freq_center: float = 8000.0
    lockingbw: float = 10500.0
    fb: float = 10500.0
    Fs: float = 48000.0
    signalthreshold: float = 0.65

    def to_c(self) -> capi.Settings:
        return capi.Settings(capi.KIND_OQPSK, self.coarsefreqest_fft_power, self.freq_center, self.lockingbw, self.fb,
                             self.Fs, self.signalthreshold)


@dataclass
class MskSettings:
    """MskDemodulator::Settings defaults (JAERO/mskdemodulator.h:24-45)."""

    coarsefreqest_fft_power: int = 13
    freq_center: float = 1000.0
    lockingbw: float = 900.0
    fb: float = 600.0
    Fs: float = 48000.0
    signalthreshold: float = 0.5

    def to_c(self) -> capi.Settings:
        return capi.Settings(capi.KIND_MSK, self.coarsefreqest_fft_power, self.freq_center, self.lockingbw, self.fb,
                             self.Fs, self.signalthreshold)


@dataclass
class BurstOqpskSettings:
    """BurstOqpskDemodulator::Settings defaults (JAERO/burstoqpskdemodulator.h:24-46)."""

    coarsefreqest_fft_power: int = 13
    freq_center: float = 8000.0
    lockingbw: float = 10500.0
    fb: float = 10500.0
    Fs: float = 48000.0
    signalthreshold: float = 0.6

    def to_c(self) -> capi.Settings:
        return capi.Settings(capi.KIND_BURST_OQPSK, self.coarsefreqest_fft_power, self.freq_center, self.lockingbw, self.fb,
                             self.Fs, self.signalthreshold)


@dataclass
class BurstMskSettings:
    """BurstMskDemodulator::Settings as the main window sets them for 600 / 1200 bps (JAERO/burstmskdemodulator.h:27-50)."""

    coarsefreqest_fft_power: int = 13
    freq_center: float = 1000.0
    lockingbw: float = 1800.0
    fb: float = 1200.0
    Fs: float = 48000.0
    signalthreshold: float = 0.6

    def to_c(self) -> capi.Settings:
        return capi.Settings(capi.KIND_BURST_MSK, self.coarsefreqest_fft_power, self.freq_center, self.lockingbw, self.fb,
                             self.Fs, self.signalthreshold)


class DemodulatorBank:
    """A bank of `nchannels` demodulators on one GPU (thin wrapper over jaero_ctx)."""

    def __init__(self, settings, nchannels: Optional[int] = None, device: int = 0, *, ebno: bool = True,
                 status_log: bool = False, capture_symbols: bool = False, trace: bool = False,
                 max_write_samples: int = 65536, softbit_capacity: int = 0):
        self.L = capi.lib()
        if isinstance(settings, (list, tuple)):
            arr = (capi.Settings * len(settings))(*[s.to_c() for s in settings])
            nch = len(settings) if nchannels is None else nchannels
            stride = C.sizeof(capi.Settings)
        else:
            arr = (capi.Settings * 1)(settings.to_c())
            nch = 1 if nchannels is None else nchannels
            stride = 0
        flags = (capi.FLAG_EBNO if ebno else 0) | (capi.FLAG_STATUS_LOG if status_log else 0) | (
            capi.FLAG_CAPTURE_SYMBOLS if capture_symbols else 0) | (capi.FLAG_TRACE if trace else 0)
        h = C.c_void_p()
        capi.check(self.L.jaero_create(device, nch, C.cast(arr, C.c_void_p), stride, flags, max_write_samples,
                                       softbit_capacity, C.byref(h)))
        self.h = h
        self.nch = nch
        self.device = device
        self.max_write_samples = max_write_samples
        self.kind = arr[0].kind
        self.fb = arr[0].fb
        self.Fs = arr[0].Fs

    def close(self):
        if getattr(self, "h", None):
            self.L.jaero_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- control surface (channel=-1: all) ----
    def set_settings(self, settings, channel: int = -1):
        """setSettings on live channels.  A change of fb / Fs / FFT power (whole bank only) re-creates the bank behind the handle with the
        state the reference's setSettings keeps (jaero_hip.h)."""
        s = settings.to_c()
        capi.check(self.L.jaero_set_settings(self.h, channel, C.byref(s)))
        if channel < 0 or self.nch == 1:
            self.fb, self.Fs = s.fb, s.Fs

    def set_flags(self, afc=False, sql=False, cpu_reduce=False, channel: int = -1):
        capi.check(self.L.jaero_set_flags(self.h, channel, int(afc), int(sql), int(cpu_reduce)))

    def set_dcd(self, dcd: bool, channel: int = -1):
        capi.check(self.L.jaero_set_dcd(self.h, channel, int(dcd)))

    def center_freq_changed(self, hz: float, channel: int = -1):
        capi.check(self.L.jaero_center_freq_changed(self.h, channel, float(hz)))

    # ---- data ----
    def write(self, pcm, layout: int = capi.PCM_CHANNEL_MAJOR, stream: int = 0):
        """pcm: numpy int16 array ([nch, n] channel-major or [n, nch] frame-major) or a torch int16 tensor on the
        bank's device with the same shapes."""
        if isinstance(pcm, np.ndarray):
            a = np.ascontiguousarray(pcm, dtype=np.int16)
            if a.ndim == 1:
                a = a.reshape(1, -1) if layout == capi.PCM_CHANNEL_MAJOR else a.reshape(-1, 1)
            n = a.shape[1] if layout == capi.PCM_CHANNEL_MAJOR else a.shape[0]
            nch = a.shape[0] if layout == capi.PCM_CHANNEL_MAJOR else a.shape[1]
            assert nch == self.nch, (a.shape, self.nch)
            capi.check(self.L.jaero_write(self.h, a.ctypes.data, n, layout, 0, stream))
        else:  # torch tensor
            t = pcm
            assert t.is_cuda and t.is_contiguous() and t.element_size() == 2
            n = t.shape[1] if layout == capi.PCM_CHANNEL_MAJOR else t.shape[0]
            nch = t.shape[0] if layout == capi.PCM_CHANNEL_MAJOR else t.shape[1]
            assert nch == self.nch, (tuple(t.shape), self.nch)
            capi.check(self.L.jaero_write(self.h, t.data_ptr(), n, layout, 1, stream))

    def read_softbits(self, channel: int, cap: int = 1 << 20) -> np.ndarray:
        buf = np.empty(cap, dtype=np.int16)
        n = C.c_int(0)
        capi.check(self.L.jaero_read_softbits(self.h, channel, buf.ctypes.data, cap, C.byref(n)))
        return buf[: n.value].copy()

    def read_softbits_all(self, cap_per_channel: int):
        buf = np.empty((self.nch, cap_per_channel), dtype=np.int16)
        counts = np.zeros(self.nch, dtype=np.int32)
        capi.check(self.L.jaero_read_softbits_all(self.h, buf.ctypes.data, cap_per_channel, counts.ctypes.data))
        return buf, counts

    def discard_softbits(self, stream: int = 0):
        capi.check(self.L.jaero_discard_softbits(self.h, stream))

    def softbits_view(self):
        p, c, cap = C.c_void_p(), C.c_void_p(), C.c_int()
        capi.check(self.L.jaero_softbits_view(self.h, C.byref(p), C.byref(c), C.byref(cap)))
        return p.value, c.value, cap.value

    def read_status(self, channel: int) -> capi.Status:
        st = capi.Status()
        capi.check(self.L.jaero_read_status(self.h, channel, C.byref(st)))
        return st

    def read_status_log(self, channel: int, caprows: int = 4096) -> np.ndarray:
        buf = np.empty((caprows, 6), dtype=np.float64)
        n = C.c_int(0)
        capi.check(self.L.jaero_read_status_log(self.h, channel, buf.ctypes.data, caprows, C.byref(n)))
        return buf[: n.value].copy()

    def read_symbols(self, channel: int, caprows: int = 1 << 18) -> np.ndarray:
        buf = np.empty((caprows, 3), dtype=np.float64)
        n = C.c_int(0)
        capi.check(self.L.jaero_read_symbols(self.h, channel, buf.ctypes.data, caprows, C.byref(n)))
        return buf[: n.value].copy()

    def read_events(self, channel: int, caprows: int = 8192) -> np.ndarray:
        """Burst banks: rows [absolute sample index, capi.EV_* kind, value] (SignalStatus / EbNo / Plottables emissions)."""
        buf = np.empty((caprows, 3), dtype=np.float64)
        n = C.c_int(0)
        capi.check(self.L.jaero_read_events(self.h, channel, buf.ctypes.data, caprows, C.byref(n)))
        return buf[: n.value].copy()

    # ---- profiling ----
    def profile_enable(self, on: bool = True):
        capi.check(self.L.jaero_profile_enable(self.h, int(on)))

    def profile_read(self, which: int, reset: bool = False):
        ms, n = C.c_double(0), C.c_int(0)
        capi.check(self.L.jaero_profile_read(self.h, which, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def profile_kernel(self, which: int) -> str:
        """Name (up to its template arguments) of the kernel this bank launches for class `which` (jaero_profile_kernel)."""
        buf = C.create_string_buffer(96)
        capi.check(self.L.jaero_profile_kernel(self.h, which, buf, 96))
        return buf.value.decode()


class Ingest:
    """Batched ingest in front of a DemodulatorBank (jaero_ingest_*): dataReceived(audio, sampleRate) per channel
    (JAERO/zmq_audioreceiver.cpp:40-79 -> oqpskdemodulator.cpp:686-693), pump() turns what all channels have in common
    into bank writes of `chunk_samples`."""

    def __init__(self, bank: DemodulatorBank, chunk_samples: int, capacity_samples: int = 0):
        self.L = bank.L
        self.bank = bank
        h = C.c_void_p()
        capi.check(self.L.jaero_ingest_create(bank.h, chunk_samples, capacity_samples, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.jaero_ingest_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dataReceived(self, channel: int, audio, sampleRate: int = 48000) -> int:
        """audio: bytes (little-endian int16 mono).  Returns 0 or capi.W_RATE; raises on overflow / unsupported rate."""
        buf = bytes(audio) if not isinstance(audio, bytes) else audio
        rc = self.L.jaero_ingest_push(self.h, channel, buf, len(buf), sampleRate)
        if rc < 0:
            capi.check(rc)
        return rc

    def queued(self, channel: int = -1) -> int:
        return self.L.jaero_ingest_queued(self.h, channel)

    def pump(self, flush: bool = False, stream: int = 0) -> int:
        n = C.c_int(0)
        capi.check(self.L.jaero_ingest_pump(self.h, 1 if flush else 0, C.c_void_p(stream), C.byref(n)))
        return n.value

    def stats(self):
        a = np.zeros(3, np.int64)
        capi.check(self.L.jaero_ingest_stats(self.h, a.ctypes.data_as(C.c_void_p)))
        return {"rate_warnings": int(a[0]), "refused_samples": int(a[1]), "samples_written": int(a[2])}


class _SingleChannelDemodulator:
    """QIODevice-shaped single-channel demodulator (one-channel bank)."""

    Settings = None
    _group = 32

    def __init__(self, parent=None, device: int = 0, **bank_kw):
        self._device = device
        self._bank_kw = bank_kw
        self._bank: Optional[DemodulatorBank] = None
        self._settings = self.Settings()
        self._afc = self._sql = self._cpu = False
        self._dcd = False
        self._open = False
        self._pending: List[int] = []
        self.processDemodulatedSoftBits: Optional[Callable[[List[int]], None]] = None
        self.SignalStatus: Optional[Callable[[bool], None]] = None
        self.MSESignal: Optional[Callable[[float], None]] = None
        self.EbNoMeasurmentSignal: Optional[Callable[[float], None]] = None
        self.Plottables: Optional[Callable[[float, float, float], None]] = None
        self._nest = 0

    # reference method names
    def setSettings(self, settings):
        self._settings = settings
        if self._bank is None:
            self._bank = DemodulatorBank(settings, 1, self._device, status_log=True, **self._bank_kw)
            self._bank.set_flags(self._afc, self._sql, self._cpu)
            self._bank.set_dcd(self._dcd)
        else:
            self._bank.set_settings(settings, 0)

    def setAFC(self, state: bool):
        self._afc = bool(state)
        if self._bank:
            self._bank.set_flags(self._afc, self._sql, self._cpu)

    def setSQL(self, state: bool):
        self._sql = bool(state)
        if self._bank:
            self._bank.set_flags(self._afc, self._sql, self._cpu)

    def setCPUReduce(self, state: bool):
        self._cpu = bool(state)
        if self._bank:
            self._bank.set_flags(self._afc, self._sql, self._cpu)

    def DCDstatSlot(self, dcd: bool):
        self._dcd = bool(dcd)
        if self._bank:
            self._bank.set_dcd(self._dcd)

    def CenterFreqChangedSlot(self, freq_center: float):
        if self._bank:
            self._bank.center_freq_changed(freq_center, 0)

    def start(self):
        if self._bank is None:
            self.setSettings(self._settings)
        self._open = True

    def stop(self):
        self._open = False

    def getCurrentFreq(self) -> float:
        return self._bank.read_status(0).freq_center

    def writeData(self, data, length: Optional[int] = None) -> int:
        """data: bytes of little-endian int16 mono PCM (as QIODevice::writeData) or an int16 numpy array."""
        if isinstance(data, (bytes, bytearray, memoryview)):
            pcm = np.frombuffer(data, dtype="<i2")
            if length is not None:
                pcm = pcm[: length // 2]
        else:
            pcm = np.asarray(data, dtype=np.int16)
        ret = 2 * pcm.shape[0]
        if pcm.shape[0] == 0:
            return 0
        step = self._bank.max_write_samples
        for s in range(0, pcm.shape[0], step):
            self._bank.write(pcm[s:s + step].reshape(1, -1))
            self._emit()
        return ret

    def _emit(self):
        soft = self._bank.read_softbits(0)
        if soft.size:
            self._pending.extend(int(v) for v in soft)
        g = self._group
        while len(self._pending) >= g:
            chunk, self._pending = self._pending[:g], self._pending[g:]
            if self.processDemodulatedSoftBits:
                self.processDemodulatedSoftBits(chunk)
        for row in self._bank.read_status_log(0):
            if self.Plottables:
                self.Plottables(row[1], row[2], self._settings.lockingbw)
            if self.EbNoMeasurmentSignal:
                self.EbNoMeasurmentSignal(row[4])
            if self.MSESignal:
                self.MSESignal(row[3])
            if self.SignalStatus:
                self.SignalStatus(bool(row[5]))


class OqpskDemodulator(_SingleChannelDemodulator):
    """Drop-in shaped like JAERO's OqpskDemodulator (JAERO/oqpskdemodulator.h:15-152); emits 32 soft bits at a time."""

    Settings = OqpskSettings
    _group = 32


class MskDemodulator(_SingleChannelDemodulator):
    """Drop-in shaped like JAERO's MskDemodulator (JAERO/mskdemodulator.h:19-168); emits 12 soft bits at a time."""

    Settings = MskSettings
    _group = 12


class _SingleChannelBurstDemodulator(_SingleChannelDemodulator):
    """Burst classes: the soft-bit stream carries -1 start-of-burst markers and is re-emitted in the reference's groups
    (first group of a burst = marker + 32 / 12 soft bits); SignalStatus / EbNoMeasurmentSignal / Plottables come from the
    event log."""

    def setSettings(self, settings):
        self._settings = settings
        if self._bank is not None:
            self._bank.close()
        self._bank = DemodulatorBank(settings, 1, self._device, **self._bank_kw)
        self._bank.set_flags(self._afc, self._sql, self._cpu)
        self._bank.set_dcd(self._dcd)

    def getCurrentFreq(self) -> float:
        return self._bank.read_status(0).freq_est

    def _emit(self):
        soft = self._bank.read_softbits(0)  # emitted groups only; the pending tail stays on the device
        # the emitted stream is a concatenation of RxDataBits emissions: a group ends when, after a pair of soft bits
        # was pushed, it holds >= 32 (12) entries; the -1 marker is pushed without that test
        g, i, n = self._group, 0, int(soft.size)
        while i < n:
            j, size = i, 0
            while j < n:
                if soft[j] == -1:
                    j, size = j + 1, size + 1
                    continue
                j, size = j + 2, size + 2
                if size >= g:
                    break
            if self.processDemodulatedSoftBits:
                self.processDemodulatedSoftBits([int(v) for v in soft[i:j]])
            i = j
        for row in self._bank.read_events(0):
            kind = int(row[1])
            if kind == capi.EV_SIGNAL and self.SignalStatus:
                self.SignalStatus(bool(row[2]))
            elif kind == capi.EV_EBNO and self.EbNoMeasurmentSignal:
                self.EbNoMeasurmentSignal(row[2])
            elif kind == capi.EV_FREQ and self.Plottables:
                self.Plottables(row[2], row[2], self._settings.lockingbw)


class BurstOqpskDemodulator(_SingleChannelBurstDemodulator):
    """Drop-in shaped like JAERO's BurstOqpskDemodulator (JAERO/burstoqpskdemodulator.h:19-228)."""

    Settings = BurstOqpskSettings
    _group = 32


class BurstMskDemodulator(_SingleChannelBurstDemodulator):
    """Drop-in shaped like JAERO's BurstMskDemodulator (JAERO/burstmskdemodulator.h:22-219)."""

    Settings = BurstMskSettings
    _group = 12


class AeroLBank:
    """A bank of AeroL bit pipelines (continuous P-channel path of JAERO/aerol.cpp AeroL::Decode): soft bits in, CRC-checked
    12-byte signal units out.  Thin wrapper over jaero_aerol_ctx."""

    def __init__(self, nchannels: int, fb: int, device: int = 0, max_softbits_per_write: int = 1 << 16, su_capacity: int = 0,
                 burst: bool = False):
        """burst=True: AeroL::setSettings(fb, burstmode=true), the R/T channel packet search behind a burst demodulator bank
        (read_packets instead of read_sus)."""
        self.L = capi.lib()
        h = C.c_void_p()
        create = self.L.jaero_aerol_create_burst if burst else self.L.jaero_aerol_create
        capi.check(create(device, nchannels, int(fb), max_softbits_per_write, su_capacity, C.byref(h)))
        self.h, self.nch, self.fb, self.burst = h, nchannels, int(fb), burst

    def close(self):
        if getattr(self, "h", None):
            self.L.jaero_aerol_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def write(self, soft: np.ndarray, counts=None, stream: int = 0):
        """soft: int16 [nch, n] (host); counts: per-channel number of valid soft bits (default: all n)."""
        a = np.ascontiguousarray(soft, dtype=np.int16)
        if a.ndim == 1:
            a = a.reshape(1, -1)
        assert a.shape[0] == self.nch
        cnt = np.full(self.nch, a.shape[1], dtype=np.int32) if counts is None else np.ascontiguousarray(counts, dtype=np.int32)
        capi.check(self.L.jaero_aerol_write(self.h, a.ctypes.data, cnt.ctypes.data, a.shape[1], int(cnt.max(initial=0)), 0, stream))

    def write_from_bank(self, bank: "DemodulatorBank", max_count: int, stream: int = 0):
        """Zero-copy: consume what a continuous DemodulatorBank has produced since its last discard (device to device)."""
        ptr, cnt, cap = bank.softbits_view()
        capi.check(self.L.jaero_aerol_write(self.h, ptr, cnt, cap, min(max_count, cap), 1, stream))
        bank.discard_softbits(stream)

    def read_sus(self, channel: int, caprows: int = 4096) -> np.ndarray:
        buf = np.empty((caprows, 16), dtype=np.int32)
        n = C.c_int(0)
        capi.check(self.L.jaero_aerol_read_sus(self.h, channel, buf.ctypes.data, caprows, C.byref(n)))
        return buf[: n.value].copy()

    def read_voice(self, channel: int, caprows: int = 256):
        """C channel (fb 8400): (frame numbers uint32[n], voice bytes uint8[n, 300]), what AeroL::DecodeC hands to Voicesignal."""
        buf = np.empty((caprows, 304), dtype=np.uint8)
        n = C.c_int(0)
        capi.check(self.L.jaero_aerol_read_voice(self.h, channel, buf.ctypes.data, caprows, C.byref(n)))
        rows = buf[: n.value]
        return rows[:, :4].copy().view(np.uint32).reshape(-1), rows[:, 4:].copy()

    def read_packets(self, channel: int, caprows: int = 4096):
        """burst mode: [(type, bytes)] with type 1 = R packet, 2 = T packet (header 6 bytes, then 12 per signal unit)."""
        buf = np.empty((caprows, 16), dtype=np.int32)
        n = C.c_int(0)
        capi.check(self.L.jaero_aerol_read_packets(self.h, channel, buf.ctypes.data, caprows, C.byref(n)))
        rows, out = buf[: n.value], []
        for pk in sorted(set(rows[:, 0].tolist())) if len(rows) else []:
            r = rows[rows[:, 0] == pk]
            r = r[np.argsort(r[:, 1])]
            out.append((int(r[0, 15]), bytes(int(v) for v in r[:, 2:14].reshape(-1))[: int(r[0, 14])]))
        return out

    def read_events(self, channel: int, caprows: int = 256) -> np.ndarray:
        buf = np.empty((caprows, 3), dtype=np.int64)
        n = C.c_int(0)
        capi.check(self.L.jaero_aerol_read_events(self.h, channel, buf.ctypes.data, caprows, C.byref(n)))
        return buf[: n.value].copy()

    def write_device(self, soft_ptr: int, counts_ptr: int, stride: int, max_count: int, stream: int = 0):
        """soft int16 [nch][stride] and counts int32 [nch] already on the bank's device."""
        capi.check(self.L.jaero_aerol_write(self.h, soft_ptr, counts_ptr, stride, max_count, 1, stream))

    def profile_enable(self, on: bool = True):
        capi.check(self.L.jaero_aerol_profile_enable(self.h, int(on)))

    def profile_read(self, which: int, reset: bool = False):
        ms, n = C.c_double(0), C.c_int(0)
        capi.check(self.L.jaero_aerol_profile_read(self.h, which, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def tick_dcd(self) -> np.ndarray:
        out = np.zeros(self.nch, dtype=np.int32)
        capi.check(self.L.jaero_aerol_tick_dcd(self.h, out.ctypes.data))
        return out
