"""ctypes binding of libjaero_hip.so (the C ABI declared in include/jaero_hip.h).

The shared library is built in-tree by `__graft_entry__.build()` / `make -C jaero_amd/csrc`.  There is no CPU or
PyTorch fallback: importing this module without the library raises, and every call fails loudly when the HIP
extension cannot run (no device, wrong architecture).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JAERO_HIP_LIB") or os.path.join(HERE, "libjaero_hip.so")  # override: A/B builds of the same ABI

KIND_MSK, KIND_OQPSK, KIND_BURST_MSK, KIND_BURST_OQPSK = 0, 1, 2, 3
FLAG_EBNO, FLAG_STATUS_LOG, FLAG_CAPTURE_SYMBOLS, FLAG_TRACE = 1, 2, 4, 8
EV_SIGNAL, EV_EBNO, EV_FREQ, EV_PEAK, EV_TRIDENT = 0, 1, 2, 3, 4
PCM_CHANNEL_MAJOR, PCM_FRAME_MAJOR = 0, 1
W_RATE = 1  # jaero_ingest_push: sample rate differs from the bank (warning, data queued)
E_OK, E_INVAL, E_NODEV, E_NOMEM, E_HIP, E_OVERFLOW, E_NOTSUP = 0, -1, -2, -3, -4, -5, -6

EXPORTS = [
    "jaero_create", "jaero_destroy", "jaero_set_settings", "jaero_set_flags", "jaero_set_dcd",
    "jaero_center_freq_changed", "jaero_write", "jaero_read_softbits", "jaero_read_softbits_all",
    "jaero_softbits_view", "jaero_discard_softbits", "jaero_read_status", "jaero_read_status_log",
    "jaero_read_symbols", "jaero_viterbi_decode_soft", "jaero_viterbi_continuous", "jaero_abi_version",
    "jaero_num_channels", "jaero_strerror", "jaero_last_error", "jaero_profile_enable", "jaero_profile_read", "jaero_profile_kernel", "jaero_debug_viterbi_layout",
    "jaero_debug_schedule", "jaero_debug_schedule_lanes", "jaero_debug_prefilter", "jaero_debug_read_prefiltered", "jaero_read_events",
    "jaero_aerol_create", "jaero_aerol_create_burst", "jaero_aerol_read_packets", "jaero_aerol_destroy", "jaero_aerol_write", "jaero_aerol_read_sus", "jaero_aerol_read_events",
    "jaero_aerol_tick_dcd", "jaero_aerol_profile_enable", "jaero_aerol_profile_read", "jaero_aerol_read_voice",
    "jaero_ingest_create", "jaero_ingest_destroy", "jaero_ingest_push", "jaero_ingest_queued", "jaero_ingest_pump",
    "jaero_ingest_stats",
    "jaero_shard_range", "jaero_comm_get_unique_id", "jaero_comm_create", "jaero_comm_destroy", "jaero_fan_out_pcm", "jaero_gather_softbits",
]


class Settings(C.Structure):
    """struct jaero_settings == OqpskDemodulator::Settings / MskDemodulator::Settings."""

    _fields_ = [
        ("kind", C.c_int),
        ("coarsefreqest_fft_power", C.c_int),
        ("freq_center", C.c_double),
        ("lockingbw", C.c_double),
        ("fb", C.c_double),
        ("Fs", C.c_double),
        ("signalthreshold", C.c_double),
    ]


class Status(C.Structure):
    _fields_ = [
        ("mse", C.c_double),
        ("ebno", C.c_double),
        ("freq_est", C.c_double),
        ("freq_center", C.c_double),
        ("signal", C.c_int),
        ("n_estimates", C.c_int),
    ]


class JaeroError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libjaero_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libjaero_hip.so; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C jaero_amd/csrc).  jaero_amd has no CPU fallback."
        )
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7.  Importing torch first makes the
    # dynamic linker resolve libjaero_hip.so's libamdhip64.so.7 dependency to that already-loaded copy, so tensors,
    # streams and this library share one runtime (loading /opt/rocm's copy first leaves torch without a device).
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover - torch is part of the image
        pass
    L = C.CDLL(LIB_PATH)
    vp, ip, dp = C.c_void_p, C.c_int, C.c_double
    L.jaero_create.argtypes = [ip, ip, vp, ip, C.c_uint, ip, ip, C.POINTER(vp)]
    L.jaero_destroy.argtypes = [vp]
    L.jaero_destroy.restype = None
    L.jaero_set_settings.argtypes = [vp, ip, C.POINTER(Settings)]
    L.jaero_set_flags.argtypes = [vp, ip, ip, ip, ip]
    L.jaero_set_dcd.argtypes = [vp, ip, ip]
    L.jaero_center_freq_changed.argtypes = [vp, ip, dp]
    L.jaero_write.argtypes = [vp, vp, ip, ip, ip, vp]
    L.jaero_read_softbits.argtypes = [vp, ip, vp, ip, C.POINTER(ip)]
    L.jaero_read_softbits_all.argtypes = [vp, vp, ip, vp]
    L.jaero_softbits_view.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(ip)]
    L.jaero_discard_softbits.argtypes = [vp, vp]
    L.jaero_read_status.argtypes = [vp, ip, C.POINTER(Status)]
    L.jaero_read_status_log.argtypes = [vp, ip, vp, ip, C.POINTER(ip)]
    L.jaero_read_symbols.argtypes = [vp, ip, vp, ip, C.POINTER(ip)]
    L.jaero_read_events.argtypes = [vp, ip, vp, ip, C.POINTER(ip)]
    L.jaero_viterbi_decode_soft.argtypes = [ip, vp, ip, ip, vp, ip, vp]
    L.jaero_viterbi_continuous.argtypes = [ip, vp, ip, ip, ip, vp, vp, vp, ip, vp]
    L.jaero_abi_version.restype = ip
    L.jaero_num_channels.argtypes = [vp]
    L.jaero_strerror.argtypes = [ip]
    L.jaero_strerror.restype = C.c_char_p
    L.jaero_last_error.restype = C.c_char_p
    L.jaero_profile_enable.argtypes = [vp, ip]
    L.jaero_profile_read.argtypes = [vp, ip, C.POINTER(dp), C.POINTER(ip), ip]
    L.jaero_profile_kernel.argtypes = [vp, ip, C.c_char_p, ip]
    L.jaero_debug_schedule.argtypes = [ip, ip, ip, vp, ip, vp, ip, C.POINTER(ip)]
    L.jaero_debug_schedule_lanes.argtypes = [ip, ip, ip, vp, vp, ip, vp, ip, vp, ip, C.POINTER(ip)]
    L.jaero_debug_prefilter.argtypes = [ip, vp, ip, dp, dp, vp]
    L.jaero_debug_read_prefiltered.argtypes = [vp, ip, vp, ip]
    L.jaero_aerol_create.argtypes = [ip, ip, ip, ip, ip, C.POINTER(vp)]
    L.jaero_aerol_create_burst.argtypes = [ip, ip, ip, ip, ip, C.POINTER(vp)]
    L.jaero_aerol_read_packets.argtypes = [vp, ip, vp, ip, C.POINTER(ip)]
    L.jaero_aerol_read_voice.argtypes = [vp, ip, vp, ip, C.POINTER(ip)]
    L.jaero_aerol_destroy.argtypes = [vp]
    L.jaero_aerol_destroy.restype = None
    L.jaero_aerol_write.argtypes = [vp, vp, vp, ip, ip, ip, vp]
    L.jaero_aerol_read_sus.argtypes = [vp, ip, vp, ip, C.POINTER(ip)]
    L.jaero_aerol_read_events.argtypes = [vp, ip, vp, ip, C.POINTER(ip)]
    L.jaero_aerol_tick_dcd.argtypes = [vp, vp]
    L.jaero_aerol_profile_enable.argtypes = [vp, ip]
    L.jaero_aerol_profile_read.argtypes = [vp, ip, C.POINTER(dp), C.POINTER(ip), ip]
    L.jaero_ingest_create.argtypes = [vp, ip, ip, C.POINTER(vp)]
    L.jaero_ingest_destroy.argtypes = [vp]
    L.jaero_ingest_destroy.restype = None
    L.jaero_ingest_push.argtypes = [vp, ip, vp, ip, C.c_uint]
    L.jaero_ingest_queued.argtypes = [vp, ip]
    L.jaero_ingest_pump.argtypes = [vp, ip, vp, C.POINTER(ip)]
    L.jaero_ingest_stats.argtypes = [vp, vp]
    L.jaero_shard_range.argtypes = [ip, ip, ip, C.POINTER(ip), C.POINTER(ip)]
    L.jaero_comm_get_unique_id.argtypes = [vp]
    L.jaero_comm_create.argtypes = [ip, ip, ip, vp, C.POINTER(vp)]
    L.jaero_comm_destroy.argtypes = [vp]
    L.jaero_comm_destroy.restype = None
    L.jaero_fan_out_pcm.argtypes = [vp, ip, vp, ip, ip, vp, vp]
    L.jaero_gather_softbits.argtypes = [vp, ip, vp, vp, ip, ip, vp, vp, vp]
    for name in EXPORTS:
        getattr(L, name)  # raises AttributeError if a declared symbol is not exported
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        L = lib()
        msg = L.jaero_last_error().decode() or L.jaero_strerror(rc).decode()
        raise JaeroError(rc, msg)
