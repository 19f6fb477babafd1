"""ReceiveContext — the batched receive path on one MI355X, over the C-ABI.

PyTorch is plumbing here: it owns device memory and the HIP stream; every
computation is a hand-written gfx950 kernel reached through libultra_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import check, ultra_hip_config, ultra_hip_geometry
from .types import Entry, LDPC_BLOCK_SIZE, ModemConfig, getBitsPerSymbol


def make_c_config(config: ModemConfig, *, entry: Entry = Entry.SYNCED, n_data_symbols: Optional[int] = None,
                  training_symbols: int = 2, max_iterations: int = 50) -> ultra_hip_config:
    c = ultra_hip_config()
    c.sample_rate, c.center_freq = config.sample_rate, config.center_freq
    c.fft_size, c.num_carriers = config.fft_size, config.num_carriers
    c.cp_mode, c.symbol_guard = int(config.cp_mode), config.symbol_guard
    c.pilot_spacing, c.use_pilots = config.pilot_spacing, int(bool(config.use_pilots))
    c.modulation, c.code_rate = int(config.modulation), int(config.code_rate)
    c.max_iterations = max_iterations
    c.entry = int(entry)
    c.training_symbols = training_symbols if Entry(entry) == Entry.PRESYNCED else 0
    if n_data_symbols is None:
        # symbols that carry one 648-bit codeword (OFDMNvisWaveform::getMinSamplesForFrame,
        # src/waveform/ofdm_cox_waveform.cpp:231-258)
        bps = config.getDataCarriers() * getBitsPerSymbol(config.modulation)
        n_data_symbols = -(-LDPC_BLOCK_SIZE // bps)
    c.n_data_symbols = n_data_symbols
    c.adaptive_eq_enabled, c.adaptive_eq_use_rls = int(bool(config.adaptive_eq_enabled)), int(bool(config.adaptive_eq_use_rls))
    c.decision_directed, c.lms_mu, c.rls_lambda = int(bool(config.decision_directed)), config.lms_mu, config.rls_lambda
    c.sync_threshold = getattr(config, "sync_threshold", 0.80)
    return c


def geometry_for(cfg: ultra_hip_config) -> ultra_hip_geometry:
    g = ultra_hip_geometry()
    check(_lib.lib().ultra_hip_geometry_for(C.byref(cfg), C.byref(g)), "ultra_hip_geometry_for")
    return g


def _torch():
    import torch
    return torch


class ReceiveContext:
    """One ultra_hip_ctx: constant tables in HBM + launches on a HIP stream."""

    def __init__(self, config: ModemConfig, *, entry: Entry = Entry.SYNCED, n_data_symbols: Optional[int] = None,
                 training_symbols: int = 2, max_iterations: int = 50, device: Optional[int] = None):
        torch = _torch()
        self.config = config
        self.cfg = make_c_config(config, entry=entry, n_data_symbols=n_data_symbols,
                                 training_symbols=training_symbols, max_iterations=max_iterations)
        self.lib = _lib.lib()
        if not torch.cuda.is_available():
            raise _lib.UltraHipError(-3, "ReceiveContext needs a HIP device (no CPU fallback exists)")
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
        self._stream = stream
        self._ctx = C.c_void_p()
        check(self.lib.ultra_hip_create(C.byref(self.cfg), self.device_index, C.c_void_p(stream), C.byref(self._ctx)),
              "ultra_hip_create")
        self.geometry = ultra_hip_geometry()
        check(self.lib.ultra_hip_get_geometry(self._ctx, C.byref(self.geometry)), "ultra_hip_get_geometry")

    # ------------------------------------------------------------------ util
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self.lib.ultra_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _dev(self, t, dtype, what):
        torch = _torch()
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.ascontiguousarray(t))
        if t.device != self.device or t.dtype != dtype or not t.is_contiguous():
            t = t.to(device=self.device, dtype=dtype).contiguous()
        return t

    @staticmethod
    def _row_stride(t):
        # torch reports stride 0 for a leading dimension of size 1
        return t.stride(0) if t.shape[0] > 1 else t.shape[1]

    def tanner_graph(self):
        g = self.geometry
        rp = np.zeros(g.ldpc_m + 1, np.uint32)
        ci = np.zeros(g.ldpc_edges, np.uint32)
        check(self.lib.ultra_hip_get_tanner_graph(self._ctx, rp.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                  ci.ctypes.data_as(C.POINTER(C.c_uint32))), "get_tanner_graph")
        return rp, ci

    def synchronize(self):
        check(self.lib.ultra_hip_synchronize(self._ctx), "ultra_hip_synchronize")

    def timer_begin(self):
        check(self.lib.ultra_hip_timer_begin(self._ctx), "ultra_hip_timer_begin")

    def timer_end(self) -> float:
        ms = C.c_float(0)
        check(self.lib.ultra_hip_timer_end(self._ctx, C.byref(ms)), "ultra_hip_timer_end")
        return float(ms.value)

    # ------------------------------------------------------------ hot path
    def _result_buffers(self, n, out):
        """dict(bytes [n][ceil(k/8)] u8, iters [n] i32, ok [n] u8): fresh, or the caller's (checked: they reach the
        kernels as raw pointers)."""
        torch = _torch()
        g = self.geometry
        if out is None:
            return dict(bytes=torch.empty((n, g.decoded_bytes), dtype=torch.uint8, device=self.device),
                        iters=torch.empty(n, dtype=torch.int32, device=self.device),
                        ok=torch.empty(n, dtype=torch.uint8, device=self.device))
        self._check_out(out["bytes"], (n, g.decoded_bytes), torch.uint8, "bytes")
        self._check_out(out["iters"], (n,), torch.int32, "iters")
        self._check_out(out["ok"], (n,), torch.uint8, "ok")
        return dict(bytes=out["bytes"], iters=out["iters"], ok=out["ok"])

    def _check_stream(self):
        """The context launches on the stream that was current when it was created; torch allocates and frees the
        tensors handed to it on whatever stream is current NOW.  A different current stream would let the caching
        allocator reuse a temporary before the queued kernels have read it, so it is refused — by EVERY method that hands
        tensor pointers to the library (generators and counters included: make_batch(cfo_hz=...) allocates and frees a
        temporary itself)."""
        torch = _torch()
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream().cuda_stream
        if cur != self._stream:
            raise _lib.UltraHipError(-1, "ReceiveContext was created on another HIP stream than torch's current one; "
                                         "create one context per stream")

    def ldpc_decode(self, llr, want_total: bool = False, out=None):
        """[n][648] f32 LLRs -> dict(bytes [n][ceil(k/8)] u8, iters [n] i32, ok [n] u8[, llr_total])."""
        torch = _torch()
        self._check_stream()
        llr = self._dev(llr, torch.float32, "llr").reshape(-1, LDPC_BLOCK_SIZE)
        n = llr.shape[0]
        out = self._result_buffers(n, out)
        total = torch.empty((n, LDPC_BLOCK_SIZE), dtype=torch.float32, device=self.device) if want_total else None
        check(self.lib.ultra_hip_ldpc_decode_batch(self._ctx, llr.data_ptr(), n, out["bytes"].data_ptr(),
                                                   out["iters"].data_ptr(), out["ok"].data_ptr(),
                                                   total.data_ptr() if want_total else None),
              "ultra_hip_ldpc_decode_batch")
        if want_total:
            out["llr_total"] = total
        return out

    def _frames(self, audio):
        torch = _torch()
        audio = self._dev(audio, torch.float32, "audio")
        if audio.dim() == 1:
            audio = audio.reshape(1, -1)
        if audio.shape[1] < self.geometry.frame_samples:
            raise _lib.UltraHipError(-1, f"audio rows hold {audio.shape[1]} samples, frame needs "
                                         f"{self.geometry.frame_samples}")
        return audio

    def _opt(self, v, n):
        if v is None:
            return None
        torch = _torch()
        v = self._dev(v, torch.float32, "per-frame scalar").reshape(-1)
        if v.numel() != n:
            raise _lib.UltraHipError(-1, "per-frame scalar array has the wrong length")
        return v

    def demod(self, audio, cfo_hz=None, cfo_phase=None, want_state: bool = False):
        """audio [n][>=frame_samples] f32 -> LLRs [n][llrs_per_frame] (+ tracker state [n][8])."""
        torch = _torch()
        self._check_stream()
        audio = self._frames(audio)
        n = audio.shape[0]
        cfo, cph = self._opt(cfo_hz, n), self._opt(cfo_phase, n)
        llr = torch.empty((n, self.geometry.llrs_per_frame), dtype=torch.float32, device=self.device)
        state = torch.empty((n, _lib.STATE_FLOATS), dtype=torch.float32, device=self.device) if want_state else None
        check(self.lib.ultra_hip_demod_batch(self._ctx, audio.data_ptr(), self._row_stride(audio),
                                             cfo.data_ptr() if cfo is not None else None,
                                             cph.data_ptr() if cph is not None else None, n, llr.data_ptr(),
                                             state.data_ptr() if want_state else None), "ultra_hip_demod_batch")
        return (llr, state) if want_state else llr

    def demod_stream(self, audio, first_symbol: int, n_symbols: int, cfo_hz=None, cfo_phase=None, want_state: bool = False,
                     want_equalized: bool = False):
        """Symbols [first_symbol, first_symbol + n_symbols) of every frame, continuing from the tracker the previous call on
        this context left behind (ultra_hip_demod_stream_batch; first_symbol == 0 starts afresh from cfo_hz / cfo_phase).
        audio rows start AT symbol first_symbol.  Returns the LLRs of the data symbols among them [n][n_data * llrs_per_symbol]
        (+ the tracker after the last symbol [n][8]) (+ with want_equalized the equalized data carriers of every data symbol,
        complex64 [n][n_data][data carriers]: what demodulateSymbol appends to the constellation ring, demodulator.cpp:199-208 —
        ultra_hip_demod_stream_batch_eq)."""
        torch = _torch()
        self._check_stream()
        audio = self._dev(audio, torch.float32, "audio")
        if audio.dim() == 1:
            audio = audio.reshape(1, -1)
        g = self.geometry
        n = audio.shape[0]
        if audio.shape[1] < n_symbols * g.symbol_samples:
            raise _lib.UltraHipError(-1, "demod_stream: audio rows must hold n_symbols symbols")
        n_train = int(self.cfg.training_symbols)
        first_data = max(first_symbol - n_train, 0)
        n_data = max(first_symbol + n_symbols - n_train, 0) - first_data
        cfo, cph = self._opt(cfo_hz, n), self._opt(cfo_phase, n)
        llr = torch.empty((n, max(n_data, 1) * g.llrs_per_symbol), dtype=torch.float32, device=self.device)
        state = torch.empty((n, _lib.STATE_FLOATS), dtype=torch.float32, device=self.device) if want_state else None
        if want_equalized:
            eq = torch.zeros((n, max(n_data, 1), 64, 2), dtype=torch.float32, device=self.device)
            check(self.lib.ultra_hip_demod_stream_batch_eq(self._ctx, audio.data_ptr(), self._row_stride(audio),
                                                           cfo.data_ptr() if cfo is not None else None,
                                                           cph.data_ptr() if cph is not None else None, n, int(first_symbol),
                                                           int(n_symbols), llr.data_ptr(), state.data_ptr() if want_state else None,
                                                           eq.data_ptr()), "ultra_hip_demod_stream_batch_eq")
        else:
            check(self.lib.ultra_hip_demod_stream_batch(self._ctx, audio.data_ptr(), self._row_stride(audio),
                                                        cfo.data_ptr() if cfo is not None else None,
                                                        cph.data_ptr() if cph is not None else None, n, int(first_symbol), int(n_symbols),
                                                        llr.data_ptr(), state.data_ptr() if want_state else None),
                  "ultra_hip_demod_stream_batch")
        llr = llr[:, :n_data * g.llrs_per_symbol]
        out = (llr, state) if want_state else (llr,)
        if want_equalized:
            out = out + (torch.view_as_complex(eq[:, :n_data, :g.n_data_carriers, :].contiguous()),)
        return out if len(out) > 1 else out[0]

    def adopt_tracker(self, src: "ReceiveContext", n_frames: int = 1) -> None:
        """ultra_hip_stream_adopt: the tracker records of frames 0 .. n_frames - 1 of `src` (another entry of the same carrier
        layout and modulation, whose last stream calls demodulated them) become this context's — what ONE OFDMDemodulator object
        carries from a processPresynced() frame into the Schmidl-Cox frame that follows it without reset()."""
        check(self.lib.ultra_hip_stream_adopt(self._ctx, src._ctx, int(n_frames)), "ultra_hip_stream_adopt")

    def demod_into(self, audio, llr, cfo_hz=None, cfo_phase=None):
        """Demodulate into the rows of a caller-owned LLR array whose row stride may exceed llrs_per_frame
        (ultra_hip_demod_batch_strided): llr = a [n][>= llrs_per_frame] f32 view with unit column stride — e.g. a row
        range of the [.][768] array the modulations of a mode grid share."""
        torch = _torch()
        self._check_stream()
        audio = self._frames(audio)
        n = audio.shape[0]
        g = self.geometry
        if (not isinstance(llr, torch.Tensor) or llr.dtype != torch.float32 or llr.device != self.device or llr.dim() != 2
                or llr.shape[0] != n or llr.shape[1] < g.llrs_per_frame or llr.stride(1) != 1):
            raise _lib.UltraHipError(-1, "demod_into: llr must be a [n][>= llrs_per_frame] f32 view with unit column stride")
        cfo, cph = self._opt(cfo_hz, n), self._opt(cfo_phase, n)
        check(self.lib.ultra_hip_demod_batch_strided(self._ctx, audio.data_ptr(), self._row_stride(audio),
                                                     cfo.data_ptr() if cfo is not None else None,
                                                     cph.data_ptr() if cph is not None else None, n, llr.data_ptr(),
                                                     self._row_stride(llr), None), "ultra_hip_demod_batch_strided")
        return llr

    def ldpc_decode_blocks(self, llr, block_len: int, block_stride: int, n_blocks: int, out=None):
        """Decode n_blocks runs of block_len codewords that lie block_stride rows apart in a [rows][stride >= 648] f32
        array (ultra_hip_ldpc_decode_blocks) -> dense dict(bytes, iters, ok) of n_blocks * block_len rows."""
        torch = _torch()
        self._check_stream()
        if (not isinstance(llr, torch.Tensor) or llr.dtype != torch.float32 or llr.device != self.device or llr.dim() != 2
                or llr.shape[1] < LDPC_BLOCK_SIZE or llr.stride(1) != 1
                or (n_blocks - 1) * block_stride + block_len > llr.shape[0] or block_stride < block_len):
            raise _lib.UltraHipError(-1, "ldpc_decode_blocks: llr must hold every block as rows of >= 648 f32")
        n = n_blocks * block_len
        out = self._result_buffers(n, out)
        check(self.lib.ultra_hip_ldpc_decode_blocks(self._ctx, llr.data_ptr(), self._row_stride(llr), block_len, block_stride,
                                                    n_blocks, out["bytes"].data_ptr(), out["iters"].data_ptr(),
                                                    out["ok"].data_ptr()), "ultra_hip_ldpc_decode_blocks")
        return out

    def demod_stream_set_cfo(self, frame: int, cfo_hz: float):
        """OFDMDemodulator::setFrequencyOffset (demodulator.cpp:805-814) for one frame of the batch in flight, between two
        demod_stream calls: CFO and filtered CFO = cfo_hz, correction phase 0, from the next symbol on."""
        check(self.lib.ultra_hip_demod_stream_set_cfo(self._ctx, int(frame), float(cfo_hz)), "ultra_hip_demod_stream_set_cfo")

    def demod_decode(self, audio, cfo_hz=None, cfo_phase=None, want_llr: bool = False, out=None):
        """Fused receive path -> dict(bytes, iters, ok[, llr]).  `out` may hand in the result tensors (and, under "llr",
        the [n][llrs_per_frame] destination of the soft bits)."""
        torch = _torch()
        self._check_stream()
        audio = self._frames(audio)
        n = audio.shape[0]
        cfo, cph = self._opt(cfo_hz, n), self._opt(cfo_phase, n)
        g = self.geometry
        llr = None
        if out is not None and out.get("llr") is not None:      # the caller's LLR buffer: soft bits are delivered, not workspace
            llr = out["llr"]
            self._check_out(llr, (n, g.llrs_per_frame), torch.float32, "llr")
        elif want_llr:
            llr = torch.empty((n, g.llrs_per_frame), dtype=torch.float32, device=self.device)
        out = self._result_buffers(n, out)
        check(self.lib.ultra_hip_demod_decode_batch(self._ctx, audio.data_ptr(), self._row_stride(audio),
                                                    cfo.data_ptr() if cfo is not None else None,
                                                    cph.data_ptr() if cph is not None else None, n,
                                                    llr.data_ptr() if llr is not None else None, out["bytes"].data_ptr(),
                                                    out["iters"].data_ptr(), out["ok"].data_ptr()),
              "ultra_hip_demod_decode_batch")
        if llr is not None:
            out["llr"] = llr
        return out

    def acquire(self, audio, chunk: int = 960):
        """Preamble acquisition of a batch of streams [n][n_samples] (OFDMDemodulator::process in the
        SEARCHING state, each stream fed `chunk` samples per call: demodulator.cpp:461-600).
        Returns device tensors dict(found, data_start, cfo_hz, sync_offset, fed_at_sync)."""
        torch = _torch()
        self._check_stream()
        audio = self._dev(audio, torch.float32, "audio")
        if audio.dim() != 2:
            raise ValueError("audio must be [n_streams][n_samples]")
        n, ns = audio.shape
        out = dict(found=torch.zeros(n, dtype=torch.int32, device=self.device),
                   data_start=torch.zeros(n, dtype=torch.int32, device=self.device),
                   cfo_hz=torch.zeros(n, dtype=torch.float32, device=self.device),
                   sync_offset=torch.zeros(n, dtype=torch.int32, device=self.device),
                   fed_at_sync=torch.zeros(n, dtype=torch.int32, device=self.device))
        check(self.lib.ultra_hip_acquire_batch(self._ctx, audio.data_ptr(), self._row_stride(audio), ns, int(chunk), n,
                                               out["found"].data_ptr(), out["data_start"].data_ptr(),
                                               out["cfo_hz"].data_ptr(), out["sync_offset"].data_ptr(),
                                               out["fed_at_sync"].data_ptr()), "ultra_hip_acquire_batch")
        return out

    def acquire_stream(self, audio, origin: int, n_samples: int, resume, midframe: bool = False):
        """ONE process() call of the SEARCHING state for live streams (ultra_hip_acquire_stream_batch): audio [n][>= n_samples
        - origin] holds samples [origin, n_samples) of every stream, resume = device int32 [n][4] in/out ({start of rx_buffer,
        fed, noise floor bits, -}).  Returns device tensors dict(found, data_start, cfo_hz, sync_offset).
        midframe=True: the preamble check of the SYNCED state instead (ultra_hip_resync_stream_batch; resume is only read)."""
        torch = _torch()
        self._check_stream()
        audio = self._dev(audio, torch.float32, "audio")
        if audio.dim() != 2:
            raise ValueError("audio must be [n_streams][n_samples - origin]")
        n = audio.shape[0]
        self._check_out(resume, (n, 4), torch.int32, "resume")
        out = dict(found=torch.zeros(n, dtype=torch.int32, device=self.device),
                   data_start=torch.zeros(n, dtype=torch.int32, device=self.device),
                   cfo_hz=torch.zeros(n, dtype=torch.float32, device=self.device),
                   sync_offset=torch.zeros(n, dtype=torch.int32, device=self.device))
        entry = "ultra_hip_resync_stream_batch" if midframe else "ultra_hip_acquire_stream_batch"
        check(getattr(self.lib, entry)(self._ctx, audio.data_ptr(), self._row_stride(audio), int(origin), int(n_samples), n,
                                       resume.data_ptr(), out["found"].data_ptr(), out["data_start"].data_ptr(),
                                       out["cfo_hz"].data_ptr(), out["sync_offset"].data_ptr()), entry)
        return out

    def chirp_sync(self, audio, threshold: float = 0.15):
        """OFDMChirpWaveform::detectSync for a batch of buffers [n][n_samples] (dual-chirp detection,
        src/sync/chirp_sync.hpp:349-505).  Returns device tensors dict(detected, start_sample, cfo_hz,
        correlation, up_chirp_start, down_chirp_start)."""
        torch = _torch()
        self._check_stream()
        audio = self._dev(audio, torch.float32, "audio")
        if audio.dim() != 2:
            raise ValueError("audio must be [n_streams][n_samples]")
        n, ns = audio.shape
        i32 = lambda: torch.zeros(n, dtype=torch.int32, device=self.device)
        f32 = lambda: torch.zeros(n, dtype=torch.float32, device=self.device)
        out = dict(detected=i32(), start_sample=i32(), cfo_hz=f32(), correlation=f32(), up_chirp_start=i32(),
                   down_chirp_start=i32())
        check(self.lib.ultra_hip_chirp_sync_batch(self._ctx, audio.data_ptr(), self._row_stride(audio), ns, n, float(threshold),
                                                  out["detected"].data_ptr(), out["start_sample"].data_ptr(),
                                                  out["cfo_hz"].data_ptr(), out["correlation"].data_ptr(),
                                                  out["up_chirp_start"].data_ptr(), out["down_chirp_start"].data_ptr()),
              "ultra_hip_chirp_sync_batch")
        return out

    def receive(self, audio, chunk: int = 960, want_llr: bool = False):
        """Raw audio streams [n][n_samples] -> acquisition -> SYNCED demodulation from each stream's own data
        start and coarse CFO -> LDPC decode (ultra_hip_receive_batch).  Returns device tensors
        dict(bytes, iters, ok, entry, cfo_hz[, llr]); entry == -1 (0xffffffff) marks streams without a frame."""
        torch = _torch()
        self._check_stream()
        audio = self._dev(audio, torch.float32, "audio")
        if audio.dim() != 2:
            raise ValueError("audio must be [n_streams][n_samples]")
        n, ns = audio.shape
        g = self.geometry
        out = dict(bytes=torch.empty((n, g.decoded_bytes), dtype=torch.uint8, device=self.device),
                   iters=torch.empty(n, dtype=torch.int32, device=self.device),
                   ok=torch.empty(n, dtype=torch.uint8, device=self.device),
                   entry=torch.empty(n, dtype=torch.int32, device=self.device),
                   cfo_hz=torch.empty(n, dtype=torch.float32, device=self.device))
        llr = torch.empty((n, g.llrs_per_frame), dtype=torch.float32, device=self.device) if want_llr else None
        check(self.lib.ultra_hip_receive_batch(self._ctx, audio.data_ptr(), self._row_stride(audio), ns, int(chunk), n,
                                               llr.data_ptr() if want_llr else None, out["bytes"].data_ptr(),
                                               out["iters"].data_ptr(), out["ok"].data_ptr(), out["entry"].data_ptr(),
                                               out["cfo_hz"].data_ptr()), "ultra_hip_receive_batch")
        if want_llr:
            out["llr"] = llr
        return out

    def chirp_receive(self, audio, threshold: float = 0.15, want_llr: bool = False):
        """Chirp-synchronised streams [n][n_samples] -> dual-chirp detection -> PRESYNCED demodulation from each
        stream's training start with the chirp CFO and its accumulated phase -> LDPC decode
        (ultra_hip_chirp_receive_batch).  Same result dict as receive()."""
        torch = _torch()
        self._check_stream()
        audio = self._dev(audio, torch.float32, "audio")
        if audio.dim() != 2:
            raise ValueError("audio must be [n_streams][n_samples]")
        n, ns = audio.shape
        g = self.geometry
        out = dict(bytes=torch.empty((n, g.decoded_bytes), dtype=torch.uint8, device=self.device),
                   iters=torch.empty(n, dtype=torch.int32, device=self.device),
                   ok=torch.empty(n, dtype=torch.uint8, device=self.device),
                   entry=torch.empty(n, dtype=torch.int32, device=self.device),
                   cfo_hz=torch.empty(n, dtype=torch.float32, device=self.device))
        llr = torch.empty((n, g.llrs_per_frame), dtype=torch.float32, device=self.device) if want_llr else None
        check(self.lib.ultra_hip_chirp_receive_batch(self._ctx, audio.data_ptr(), self._row_stride(audio), ns, n,
                                                     float(threshold), llr.data_ptr() if want_llr else None,
                                                     out["bytes"].data_ptr(), out["iters"].data_ptr(),
                                                     out["ok"].data_ptr(), out["entry"].data_ptr(),
                                                     out["cfo_hz"].data_ptr()), "ultra_hip_chirp_receive_batch")
        if want_llr:
            out["llr"] = llr
        return out

    def decode_frames(self, soft):
        """v2 frames from their soft bits [n_frames][n_soft] (RxPipeline::processFrame behind getSoftBits:
        ping check, per-codeword deinterleave if set_deinterleave() is on, CW0 -> header -> remaining codewords ->
        reassembly; ultra_hip_decode_frames_batch).  Returns device tensors dict(results [n][8] int32 =
        success, is_ping, frame_type, codewords_ok, codewords_failed, expected_codewords, frame_len, status;
        frame_data [n][stride] uint8)."""
        torch = _torch()
        self._check_stream()
        soft = self._dev(soft, torch.float32, "soft")
        if soft.dim() != 2:
            raise ValueError("soft must be [n_frames][n_soft]")
        n, ns = soft.shape
        stride = max((ns // 648) * (self.geometry.ldpc_k // 8), 1)
        out = dict(results=torch.zeros((n, 8), dtype=torch.int32, device=self.device),
                   frame_data=torch.zeros((n, stride), dtype=torch.uint8, device=self.device))
        check(self.lib.ultra_hip_decode_frames_batch(self._ctx, soft.data_ptr(), self._row_stride(soft), ns, n,
                                                     out["results"].data_ptr(), out["frame_data"].data_ptr(), stride),
              "ultra_hip_decode_frames_batch")
        return out

    def make_batch(self, n_frames: int, seed: int = 0x5EED, first_frame: int = 0, channel: str = "awgn",
                   snr_db: float = 30.0, delay_ms: float = 0.5, doppler_hz: float = 0.1, out=None, cfo_hz: float = 0.0):
        """Synthetic frames at the SYNCED entry, generated on the device (ultra_hip_make_batch): random
        payload -> encode -> preamble + modulate -> 0.5 peak -> channel.  Returns (audio [n][frame_samples],
        payload [n][k // 8]) device tensors; `out` may hand in that pair (e.g. row slices of a larger batch) to be
        overwritten.  Bit-identical to the oracle's uo_make_batch for channel "none"."""
        torch = _torch()
        self._check_stream()
        g = self.geometry
        if out is None:
            audio = torch.empty((n_frames, g.frame_samples), dtype=torch.float32, device=self.device)
            payload = torch.empty((n_frames, g.ldpc_k // 8), dtype=torch.uint8, device=self.device)
        else:
            audio, payload = out
            self._check_out(audio, (n_frames, g.frame_samples), torch.float32, "audio")
            self._check_out(payload, (n_frames, g.ldpc_k // 8), torch.uint8, "payload")
        kind = dict(none=0, awgn=1, watterson=2)[channel]
        check(self.lib.ultra_hip_make_batch(self._ctx, int(seed), int(first_frame), n_frames, kind, float(snr_db),
                                            float(delay_ms), float(doppler_hz), audio.data_ptr(), self._row_stride(audio),
                                            payload.data_ptr()), "ultra_hip_make_batch")
        if abs(float(cfo_hz)) > 0.001:           # WattersonChannel::process shifts after the noise (hf_channel.hpp:161-165)
            shifted = self.channel_cfo(audio, cfo_hz)
            if out is None:
                audio = shifted
            else:
                audio.copy_(shifted)
        return audio, payload

    def reserve(self, n_frames: int):
        """Size the context's workspaces for batches of up to n_frames (otherwise they grow inside the first call that needs
        more, with a stream synchronisation and an allocation)."""
        check(self.lib.ultra_hip_reserve(self._ctx, int(n_frames)), "ultra_hip_reserve")

    def status(self) -> dict:
        """ultra_hip_get_status: which of the context's own fall-back paths it has taken since creation (or clear_status) —
        `paths`: names of the ULTRA_HIP_ST_* bits set, empty on the default path — and what the decoder's screen decided on the
        last launch that sampled.  Synchronises the stream."""
        st = _lib.ultra_hip_path_status()
        check(self.lib.ultra_hip_get_status(self._ctx, C.byref(st)), "ultra_hip_get_status")
        out = {n: int(getattr(st, n)) for n, _ in st._fields_ if n != "reserved"}
        out["paths"] = [name for bit, name in sorted(_lib.STATUS_FLAGS.items()) if st.flags & bit]
        return out

    def clear_status(self):
        check(self.lib.ultra_hip_clear_status(self._ctx), "ultra_hip_clear_status")

    def set_workspace_limit(self, n_bytes: int):
        """Cap each per-(frame, symbol) demodulator workspace (ultra_hip_set_workspace_limit); 0 = none."""
        check(self.lib.ultra_hip_set_workspace_limit(self._ctx, int(n_bytes)), "ultra_hip_set_workspace_limit")

    def channel_cfo(self, audio, cfo_hz: float):
        """The channel's carrier frequency offset (WattersonChannel::applyCFO, hf_channel.hpp:161-232; every row by a fresh
        channel) applied to a batch of audio rows -> new tensor.  Bit-identical to the reference."""
        torch = _torch()
        self._check_stream()
        audio = self._dev(audio, torch.float32, "audio")
        if audio.dim() != 2:
            raise _lib.UltraHipError(-1, "channel_cfo: audio must be [n_frames][n_samples]")
        out = torch.empty_like(audio)
        check(self.lib.ultra_hip_channel_cfo_batch(self._ctx, audio.data_ptr(), self._row_stride(audio), out.data_ptr(),
                                                   self._row_stride(out), audio.shape[1], audio.shape[0], float(cfo_hz)),
              "ultra_hip_channel_cfo_batch")
        return out

    def make_raw_batch(self, n_streams: int, seed: int = 0x5EED, first_frame: int = 0, channel: str = "awgn",
                       snr_db: float = 30.0, lead: int = 1120, tail: int = 960, delay_ms: float = 0.5, doppler_hz: float = 0.1):
        """Raw-audio streams for receive(): [lead silence][preamble][data symbols][tail silence], 0.5 peak, AWGN on every
        sample or ("watterson") the transmission through the two-path fading channel of make_batch with its noise on every
        sample (ultra_hip_make_raw_batch_channel).  Returns (audio [n][lead + preamble + frame_samples + tail], payload)."""
        torch = _torch()
        self._check_stream()
        g = self.geometry
        n_out = lead + 7 * (self.config.fft_size + g.cp_len) + g.frame_samples + tail
        audio = torch.empty((n_streams, n_out), dtype=torch.float32, device=self.device)
        payload = torch.empty((n_streams, g.ldpc_k // 8), dtype=torch.uint8, device=self.device)
        kind = dict(none=0, awgn=1, watterson=2)[channel]
        check(self.lib.ultra_hip_make_raw_batch_channel(self._ctx, int(seed), int(first_frame), n_streams, kind, float(snr_db),
                                                        float(delay_ms), float(doppler_hz), int(lead), int(tail), audio.data_ptr(),
                                                        self._row_stride(audio), payload.data_ptr()),
              "ultra_hip_make_raw_batch_channel")
        return audio, payload

    def make_llr_batch(self, n_cw: int, esn0_db: float, seed: int = 0x5EED, first_cw: int = 0, out=None):
        """BPSK-over-AWGN LLRs of n_cw random codewords of the context's rate, generated on the device
        (ultra_hip_make_llr_batch).  Returns (llr [n][648] f32, payload [n][k // 8] u8); `out` may
        hand in that pair to be overwritten.  Bit-identical to the oracle's uo_make_llr_batch."""
        torch = _torch()
        self._check_stream()
        g = self.geometry
        if out is None:
            llr = torch.empty((n_cw, LDPC_BLOCK_SIZE), dtype=torch.float32, device=self.device)
            payload = torch.empty((n_cw, g.ldpc_k // 8), dtype=torch.uint8, device=self.device)
        else:
            llr, payload = out
            self._check_out(llr, (n_cw, LDPC_BLOCK_SIZE), torch.float32, "llr")
            self._check_out(payload, (n_cw, g.ldpc_k // 8), torch.uint8, "payload")
        check(self.lib.ultra_hip_make_llr_batch(self._ctx, int(seed), int(first_cw), n_cw, float(esn0_db), llr.data_ptr(),
                                                payload.data_ptr()), "ultra_hip_make_llr_batch")
        return llr, payload

    def _check_out(self, t, shape, dtype, what):
        """A caller-supplied output tensor goes to the kernels as a raw pointer: refuse anything but the exact
        shape / dtype / device, contiguous."""
        torch = _torch()
        if (not isinstance(t, torch.Tensor) or tuple(t.shape) != tuple(shape) or t.dtype != dtype
                or t.device != self.device or not t.is_contiguous()):
            raise _lib.UltraHipError(-1, f"output tensor '{what}' must be a contiguous {dtype} tensor of shape "
                                         f"{tuple(shape)} on {self.device}")

    def set_deinterleave(self, bits_per_symbol: int):
        """RxPipeline::setInterleaverConfig + deinterleaveCodewords (rx_pipeline.cpp:24-31,475-491): every
        codeword is passed through ChannelInterleaver(bits_per_symbol, 648)::deinterleave before it is
        decoded, fused into the decoder's LLR load.  0 switches it off."""
        check(self.lib.ultra_hip_set_deinterleave(self._ctx, int(bits_per_symbol)), "ultra_hip_set_deinterleave")

    def set_deinterleave_table(self, index):
        """Fuse an arbitrary permutation of the 648 soft bits into the decoder's LLR load: out[j] = in[index[j]]
        (ultra_hip_set_deinterleave_table; e.g. Interleaver(rows, cols).permutation).  None switches it off."""
        if index is None:
            check(self.lib.ultra_hip_set_deinterleave_table(self._ctx, None, 0), "ultra_hip_set_deinterleave_table")
            return
        idx = np.ascontiguousarray(index, dtype=np.uint16).reshape(-1)
        check(self.lib.ultra_hip_set_deinterleave_table(self._ctx, idx.ctypes.data_as(C.c_void_p), idx.size),
              "ultra_hip_set_deinterleave_table")

    KERNEL_CLASSES = ("init_state_kernel", "mix_fft_kernel", "track_kernel", "ldpc_decode_kernel", "count_errors_kernel",
                      "acquire_kernel", "chirp_sync_kernel", "track_pilot_kernel", "cfo_walk_kernel")

    def profile_enable(self, on: bool = True):
        """Bracket every kernel launch of this context with HIP events (ultra_hip_profile_enable)."""
        check(self.lib.ultra_hip_profile_enable(self._ctx, 1 if on else 0), "ultra_hip_profile_enable")

    def profile_read(self):
        """{kernel class: (total ms, launches)} of the launches recorded since the last read."""
        ms = (C.c_float * len(self.KERNEL_CLASSES))()
        cnt = (C.c_uint32 * len(self.KERNEL_CLASSES))()
        check(self.lib.ultra_hip_profile_read(self._ctx, ms, cnt), "ultra_hip_profile_read")
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(self.KERNEL_CLASSES)}

    KERNEL_CLASSES_ITEMS = KERNEL_CLASSES + ("mix_fft_rot_kernel",)

    def profile_read_items(self):
        """{kernel class: (total ms, launches, work items)} — ultra_hip_profile_read_items: the rotating transform apart from
        the instance without rotation, and the frame-symbols / codewords / streams the recorded launches covered."""
        n = len(self.KERNEL_CLASSES_ITEMS)
        ms, cnt, items = (C.c_float * n)(), (C.c_uint32 * n)(), (C.c_uint64 * n)()
        check(self.lib.ultra_hip_profile_read_items(self._ctx, ms, cnt, items), "ultra_hip_profile_read_items")
        return {k: (float(ms[i]), int(cnt[i]), int(items[i])) for i, k in enumerate(self.KERNEL_CLASSES_ITEMS)}

    def count_errors(self, result, payload, counters=None):
        """Accumulate the Monte-Carlo counters (device int64[8]) for a decoded batch."""
        torch = _torch()
        self._check_stream()
        payload = self._dev(payload, torch.uint8, "payload")
        n, pb = payload.shape
        if counters is None:
            counters = torch.zeros(8, dtype=torch.int64, device=self.device)
        check(self.lib.ultra_hip_count_errors(self._ctx, result["bytes"].data_ptr(), result["iters"].data_ptr(),
                                              result["ok"].data_ptr(), payload.data_ptr(), pb, n,
                                              counters.data_ptr()), "ultra_hip_count_errors")
        return counters

    def count_errors_points(self, result, payload, counters):
        """The counters of a batch that holds n_points sweep points of equal size back to back: rows
        [p * n, (p + 1) * n) accumulate into counters[p] (device int64 [n_points][8]); one launch."""
        torch = _torch()
        self._check_stream()
        payload = self._dev(payload, torch.uint8, "payload")
        rows, pb = payload.shape
        self._check_out(counters, (counters.shape[0], 8), torch.int64, "counters")
        n_points = counters.shape[0]
        if n_points == 0 or rows % n_points != 0 or result["iters"].shape[0] != rows:
            raise _lib.UltraHipError(-1, "count_errors_points: the batch must hold counters.shape[0] points of equal size")
        check(self.lib.ultra_hip_count_errors_points(self._ctx, result["bytes"].data_ptr(), result["iters"].data_ptr(),
                                                     result["ok"].data_ptr(), payload.data_ptr(), pb, n_points, rows // n_points,
                                                     counters.data_ptr()), "ultra_hip_count_errors_points")
        return counters

    def selftest_math(self, fn: int, a, b=None):
        torch = _torch()
        self._check_stream()
        a = self._dev(a, torch.float32, "a")
        b = self._dev(b, torch.float32, "b") if b is not None else None
        out = torch.empty_like(a)
        check(self.lib.ultra_hip_selftest_math(self._ctx, fn, a.data_ptr(), b.data_ptr() if b is not None else None,
                                               out.data_ptr(), a.numel()), "ultra_hip_selftest_math")
        return out
