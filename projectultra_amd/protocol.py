"""The v2 wire format on the receive path: mirror of the part of gui::RxPipeline that turns the soft bits of a
frame into an RxFrameResult (src/gui/modem/rx_pipeline.hpp:39-48,76-84; rx_pipeline.cpp:20-31,283-346,348-444),
running on the HIP path (ultra_hip_decode_frames_batch): all codewords of all frames in one LDPC batch, header
parsing / CRC / reassembly in a kernel behind it."""
from __future__ import annotations

from dataclasses import dataclass
from enum import IntEnum

import numpy as np

from .engine import ReceiveContext
from .types import CodeRate, LDPC_BLOCK_SIZE, ModemConfig


class FrameType(IntEnum):            # src/protocol/frame_v2.hpp:163-186
    PING = 0x01; PONG = 0x02
    PROBE = 0x10; PROBE_ACK = 0x11; CONNECT = 0x12; CONNECT_ACK = 0x13; CONNECT_NAK = 0x14; DISCONNECT = 0x15
    KEEPALIVE = 0x16; MODE_CHANGE = 0x17; ACK = 0x20; NACK = 0x21; BEACON = 0x40
    DATA = 0x30; DATA_START = 0x31; DATA_CONT = 0x32; DATA_END = 0x33


class FrameStatus(IntEnum):          # include/ultra_hip.h ultra_hip_frame_status
    CW0_FAILED = 0; BAD_HEADER = 1; WAITING = 2; CODEWORDS_FAILED = 3; COMPLETE = 4; PING = 5


@dataclass
class RxFrameResult:                 # rx_pipeline.hpp:39-48
    success: bool = False
    frame_data: bytes = b""
    frame_type: int = FrameType.PROBE
    codewords_ok: int = 0
    codewords_failed: int = 0
    snr_estimate: float = 0.0
    cfo_estimate: float = 0.0
    is_ping: bool = False


class RxFrameDecoder:
    """setDataMode / setInterleavingEnabled / setInterleaverConfig as on RxPipeline; decode_soft_bits() is the
    tail of processFrame for one frame, decode_batch() the same for many."""

    def __init__(self, device=None):
        self._device = device
        self._rate = CodeRate.R1_4
        self._connected = False
        self._interleaving = True
        # the constructor already builds ChannelInterleaver(60, 648) and interleaving is on (rx_pipeline.cpp:13-18,
        # rx_pipeline.hpp:177,182): a default decoder deinterleaves with 60 bits per symbol
        self._bits_per_symbol = 60
        self._ctx = {}
        self._expected = 0

    def setDataMode(self, rate: CodeRate, connected: bool) -> None:
        self._rate, self._connected = CodeRate(rate), bool(connected)

    def setInterleavingEnabled(self, enabled: bool) -> None:
        self._interleaving = bool(enabled)

    def setInterleaverConfig(self, bits_per_symbol: int) -> None:   # rx_pipeline.cpp:24-31
        self._bits_per_symbol = int(bits_per_symbol)

    def getExpectedCodewords(self) -> int:
        return self._expected

    def isAccumulating(self) -> bool:
        return self._expected > 0

    def _context(self) -> ReceiveContext:
        rate = self._rate if self._connected else CodeRate.R1_4          # rx_pipeline.cpp:356-366
        bps = self._bits_per_symbol if self._interleaving else 0   # deinterleaveCodewords, rx_pipeline.cpp:474-477
        key = (rate, bps)
        if key not in self._ctx:
            ctx = ReceiveContext(ModemConfig(code_rate=rate), device=self._device)
            ctx.set_deinterleave(bps)
            self._ctx[key] = ctx
        return self._ctx[key]

    def decode_batch(self, soft):
        """soft [n_frames][n_soft] -> (results [n][8] int32 numpy, list of frame bytes)."""
        out = self._context().decode_frames(soft)
        res = out["results"].cpu().numpy()
        data = out["frame_data"].cpu().numpy()
        return res, [bytes(d[:r[6]]) for d, r in zip(data, res)]

    def decode_soft_bits(self, soft_bits) -> RxFrameResult:
        soft = np.ascontiguousarray(soft_bits, np.float32).reshape(1, -1)
        r = RxFrameResult()
        if soft.size == 0:
            return r
        res, data = self.decode_batch(soft)
        res = res[0]
        r.success, r.is_ping, r.frame_type = bool(res[0]), bool(res[1]), int(res[2])
        r.codewords_ok, r.codewords_failed, r.frame_data = int(res[3]), int(res[4]), data[0]
        self._expected = int(res[5])
        return r
