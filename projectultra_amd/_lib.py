"""ctypes binding of the C-ABI in include/ultra_hip.h (libultra_hip.so).

The library holds the hand-written gfx950 kernels; there is NO CPU fallback.
If the shared object is missing this module raises at import of the symbols —
the product path fails loudly instead of silently computing somewhere else.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
# ULTRA_HIP_LIB: a VARIANT build of the same sources (tools/variants_bench.sh, tools/mix_fft_stalls.py) — never a fallback:
# whatever is named must exist and export the whole ABI
LIB_PATH = Path(os.environ["ULTRA_HIP_LIB"]).resolve() if os.environ.get("ULTRA_HIP_LIB") else PKG_DIR / "libultra_hip.so"
CSRC_DIR = PKG_DIR / "csrc"

ULTRA_HIP_ABI_VERSION = 10
STATE_FLOATS = 8


class UltraHipError(RuntimeError):
    def __init__(self, status: int, what: str):
        self.status = status
        super().__init__(f"{what}: {status_text(status)} (status {status})")


class ultra_hip_config(C.Structure):
    """POD mirror of the ModemConfig fields the path reads (include/ultra_hip.h)."""
    _fields_ = [(n, C.c_uint32) for n in (
        "sample_rate", "center_freq", "fft_size", "num_carriers", "cp_mode", "symbol_guard",
        "pilot_spacing", "use_pilots", "modulation", "code_rate", "max_iterations",
        "n_data_symbols", "entry", "training_symbols",
        "adaptive_eq_enabled", "adaptive_eq_use_rls", "decision_directed")] + [("lms_mu", C.c_float), ("rls_lambda", C.c_float), ("sync_threshold", C.c_float)]


class ultra_hip_geometry(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "cp_len", "symbol_samples", "frame_samples", "n_data_carriers", "n_pilot_carriers",
        "bits_per_carrier", "llrs_per_symbol", "llrs_per_frame", "ldpc_n", "ldpc_k", "ldpc_m",
        "ldpc_edges", "decoded_bytes")]


class ultra_hip_counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "frames", "frame_errors", "bit_errors", "info_bits", "ldpc_fail", "iters_sum",
        "undetected_errors", "reserved")]


class ultra_hip_path_status(C.Structure):
    """ultra_hip_get_status (ABI 10): fall-back paths a context has taken + the decoder screen's last decision."""
    _fields_ = [(n, C.c_uint32) for n in (
        "flags", "screen_launches", "screen_sample_n", "screen_sample_clean", "screen_gate", "screen_gate_open",
        "screen_dirty", "reserved")]


# ULTRA_HIP_ST_* (include/ultra_hip.h)
STATUS_FLAGS = {
    0x01: "demod_workspace_fallback", 0x02: "ldpc_message_kernel", 0x04: "lds_probe_failed", 0x08: "screen_list_unavailable",
    0x10: "acq_cache_unavailable", 0x20: "forced_fallback_chain", 0x40: "forced_message_kernel", 0x80: "screen_overridden",
}


COUNTER_NAMES = tuple(n for n, _ in ultra_hip_counters._fields_)

_vp, _sz, _i, _u32p = C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint32)
_cfgp, _geop = C.POINTER(ultra_hip_config), C.POINTER(ultra_hip_geometry)

# name -> (restype, argtypes); exactly the prototypes of include/ultra_hip.h
PROTOTYPES = {
    "ultra_hip_abi_version": (_i, []),
    "ultra_hip_host_sync_count": (C.c_ulonglong, []),
    "ultra_hip_strerror": (C.c_char_p, [_i]),
    "ultra_hip_device_count": (_i, []),
    "ultra_hip_geometry_for": (_i, [_cfgp, _geop]),
    "ultra_hip_create": (_i, [_cfgp, _i, _vp, C.POINTER(_vp)]),
    "ultra_hip_destroy": (None, [_vp]),
    "ultra_hip_get_geometry": (_i, [_vp, _geop]),
    "ultra_hip_get_tanner_graph": (_i, [_vp, _u32p, _u32p]),
    "ultra_hip_ldpc_decode_batch": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "ultra_hip_demod_batch": (_i, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _vp]),
    "ultra_hip_demod_batch_strided": (_i, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _sz, _vp]),
    "ultra_hip_ldpc_decode_blocks": (_i, [_vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp, _vp]),
    "ultra_hip_demod_stream_batch": (_i, [_vp, _vp, _sz, _vp, _vp, _sz, C.c_uint32, C.c_uint32, _vp, _vp]),
    "ultra_hip_stream_adopt": (_i, [_vp, _vp, _sz]),
    "ultra_hip_demod_stream_batch_eq": (_i, [_vp, _vp, _sz, _vp, _vp, _sz, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "ultra_hip_demod_stream_set_cfo": (_i, [_vp, _sz, C.c_float]),
    "ultra_hip_demod_stream_start": (_i, [_vp, _i, _vp]),
    "ultra_hip_demod_stream_set_cfo_phase": (_i, [_vp, _sz, C.c_float, C.c_float]),
    "ultra_hip_acquire_stream_batch": (_i, [_vp, _vp, _sz, C.c_uint32, C.c_uint32, _sz, _vp, _vp, _vp, _vp, _vp]),
    "ultra_hip_resync_stream_batch": (_i, [_vp, _vp, _sz, C.c_uint32, C.c_uint32, _sz, _vp, _vp, _vp, _vp, _vp]),
    "ultra_hip_demod_decode_batch": (_i, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "ultra_hip_count_errors": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _vp]),
    "ultra_hip_reserve": (_i, [_vp, _sz]),
    "ultra_hip_count_errors_points": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _vp]),
    "ultra_hip_channel_cfo_batch": (_i, [_vp, _vp, _sz, _vp, _sz, C.c_uint32, _sz, C.c_float]),
    "ultra_hip_synchronize": (_i, [_vp]),
    "ultra_hip_timer_begin": (_i, [_vp]),
    "ultra_hip_timer_end": (_i, [_vp, C.POINTER(C.c_float)]),
    "ultra_hip_acquire_batch": (_i, [_vp, _vp, _sz, C.c_uint32, C.c_uint32, _sz, _vp, _vp, _vp, _vp, _vp]),
    "ultra_hip_chirp_sync_batch": (_i, [_vp, _vp, _sz, C.c_uint32, _sz, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ultra_hip_chirp_receive_batch": (_i, [_vp, _vp, _sz, C.c_uint32, _sz, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ultra_hip_decode_frames_batch": (_i, [_vp, _vp, _sz, C.c_uint32, _sz, _vp, _vp, _sz]),
    "ultra_hip_counters_allreduce": (_i, [_vp, _vp, _vp]),
    "ultra_hip_receive_batch": (_i, [_vp, _vp, _sz, C.c_uint32, C.c_uint32, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ultra_hip_make_batch": (_i, [_vp, C.c_uint64, C.c_uint64, _sz, _i, C.c_float, C.c_float, C.c_float, _vp, _sz, _vp]),
    "ultra_hip_make_raw_batch": (_i, [_vp, C.c_uint64, C.c_uint64, _sz, _i, C.c_float, C.c_uint32, C.c_uint32, _vp, _sz, _vp]),
    "ultra_hip_make_raw_batch_channel": (_i, [_vp, C.c_uint64, C.c_uint64, _sz, _i, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32,
                                              _vp, _sz, _vp]),
    "ultra_hip_make_llr_batch": (_i, [_vp, C.c_uint64, C.c_uint64, _sz, C.c_float, _vp, _vp]),
    "ultra_hip_set_deinterleave": (_i, [_vp, C.c_uint32]),
    "ultra_hip_set_deinterleave_table": (_i, [_vp, _vp, C.c_uint32]),
    "ultra_hip_channel_interleaver_step": (_i, [C.c_uint32, C.c_uint32, _u32p]),
    "ultra_hip_get_status": (_i, [_vp, C.POINTER(ultra_hip_path_status)]),
    "ultra_hip_clear_status": (_i, [_vp]),
    "ultra_hip_set_workspace_limit": (_i, [_vp, _sz]),
    "ultra_hip_profile_enable": (_i, [_vp, _i]),
    "ultra_hip_profile_read": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "ultra_hip_profile_read_items": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "ultra_hip_malloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "ultra_hip_free": (_i, [_vp, _vp]),
    "ultra_hip_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "ultra_hip_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "ultra_hip_memcpy_h2d_async": (_i, [_vp, _vp, _vp, _sz]),
    "ultra_hip_memset": (_i, [_vp, _vp, _i, _sz]),
    "ultra_hip_host_block": (_i, [_vp, _sz, C.POINTER(_vp), C.POINTER(_vp)]),
    "ultra_hip_stage_input": (_i, [_vp, _vp, _sz, C.POINTER(_vp)]),
    "ultra_hip_stream_post": (_i, [_vp, _vp, C.c_uint32]),
    "ultra_hip_host_wait": (_i, [_vp, _vp, C.c_uint32, C.c_uint32]),
    "ultra_hip_selftest_math": (_i, [_vp, _i, _vp, _vp, _vp, _sz]),
}

_LIB = None


def source_hash() -> str:
    """sha256[:16] over the device/host sources the library is built from (csrc/*, include/ultra_hip.h), in name order.
    Evidence files that describe the KERNELS (profiles/traffic*.json, SQ counters) carry it, so a reader — bench.py — can
    tell whether they were collected on the code that is in the tree now; it works on the GPU box, where no .git exists."""
    import hashlib
    h = hashlib.sha256()
    # (the Makefile too: the compile flags decide the kernels' code as much as the sources do)
    for f in sorted(list(CSRC_DIR.glob("*.h")) + list(CSRC_DIR.glob("*.hip")) + [CSRC_DIR / "Makefile", PKG_DIR.parent / "include" / "ultra_hip.h"]):
        h.update(f.name.encode()); h.update(b"\0"); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def build(force: bool = False) -> Path:
    """Compile libultra_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [CSRC_DIR / "ultra_hip.hip", CSRC_DIR / "Makefile"] + sorted(CSRC_DIR.glob("*.h")) + [PKG_DIR.parent / "include" / "ultra_hip.h"]
    stale = (not LIB_PATH.exists()) or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", str(CSRC_DIR)] + (["-B"] if force else []))
    return LIB_PATH


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension is the product path and there is no CPU "
                f"fallback. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {CSRC_DIR}`.")
        handle = C.CDLL(str(LIB_PATH))
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)   # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if handle.ultra_hip_abi_version() != ULTRA_HIP_ABI_VERSION:
            raise ImportError("libultra_hip.so ABI version mismatch")
        _LIB = handle
    return _LIB


def status_text(status: int) -> str:
    return lib().ultra_hip_strerror(status).decode()


def check(status: int, what: str) -> None:
    if status != 0:
        raise UltraHipError(status, what)
