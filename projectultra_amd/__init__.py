"""projectultra_amd — MI355X-native batched OFDM-demodulate + LDPC-decode receive path.

A drop-in for ONE hot path of secup/ProjectUltra (the post-sync receive chain behind
IWaveform / OFDMDemodulator / LDPCDecoder), built from scratch as hand-written gfx950 HIP
kernels behind the C-ABI in include/ultra_hip.h.  This package is the host-side mirror of
the reference's interface for that path; it holds no numerics of its own.
"""
from .types import (CodeRate, CyclicPrefixMode, Entry, LDPC_BLOCK_SIZE, ModemConfig, Modulation, presets,
                    getBitsPerSymbol, getCodeRateValue, info_bits, is_differential)
from ._lib import UltraHipError, build

__all__ = ["CodeRate", "CyclicPrefixMode", "Entry", "LDPC_BLOCK_SIZE", "ModemConfig", "Modulation", "presets",
           "getBitsPerSymbol", "getCodeRateValue", "info_bits", "is_differential", "UltraHipError", "build",
           "ReceiveContext", "LDPCDecoder", "ChannelInterleaver", "Interleaver", "OFDMDemodulator", "HipOfdmWaveform", "SyncResult",
           "RxFrameDecoder", "RxFrameResult", "FrameType", "FrameStatus"]


def __getattr__(name):
    # the classes below need the HIP library at construction time; import lazily so that
    # pure-host users (geometry, config, sharding arithmetic) work without a GPU
    if name == "ReceiveContext":
        from .engine import ReceiveContext
        return ReceiveContext
    if name in ("LDPCDecoder", "ChannelInterleaver", "Interleaver"):
        from . import fec
        return getattr(fec, name)
    if name == "OFDMDemodulator":
        from .ofdm import OFDMDemodulator
        return OFDMDemodulator
    if name in ("HipOfdmWaveform", "SyncResult"):
        from . import waveform
        return getattr(waveform, name)
    if name in ("RxFrameDecoder", "RxFrameResult", "FrameType", "FrameStatus"):
        from . import protocol
        return getattr(protocol, name)
    raise AttributeError(name)
